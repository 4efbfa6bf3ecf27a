"""
Brute-force Hamming ``HashIndex`` on MI355X.

Drop-in counterpart of ``LinearHashIndex``
(smqtk_indexing/impls/hash_index/linear.py:27-244).  The reference keeps a
python ``set`` of arbitrary-precision ints and answers ``nn`` with
``heapq.nsmallest`` over ``metrics.hamming_distance`` (linear.py:235-238).
Here the index is the sorted array of UNIQUE packed codes
(``uint64[n, W]``, MSB-first, utils/bits.py) resident in HBM; ``nn`` calls
``sq_hamming_search``.  Semantics kept: build replaces, update unions, remove
raises ``KeyError`` before touching anything (linear.py:197-204), distances are
normalised by the QUERY bit length (linear.py:233,243), a code is never
returned twice, the cache element is rewritten after every mutation
(linear.py:165,182,204) and a read-only cache raises ``ValueError``
(linear.py:136-138).

Tie order: the reference's is python-set iteration order; this class returns
(distance, code value) ascending.

Cache format: ``numpy.save`` of the packed ``uint64[n, W]`` array.  The
reference's ``numpy.save(tuple(ints))`` form is read when it is a 1-D integer
array (codes < 2**63).
"""
from io import BytesIO
import threading
from typing import Any, Dict, Iterable, Optional, Set, Tuple, Type, TypeVar

import numpy as np

from .. import _require_usable
from ... import _lib
from ..._compat import (DataElement, from_config_dict, make_default_config,
                        merge_dict, to_config_dict)
from ...interfaces.hash_index import HashIndex
from ...utils.bits import (ints_to_packed, pack_bits_msb, packed_to_ints,
                           unpack_bits_msb, words_for_bits)

T = TypeVar("T", bound="HipLinearHashIndex")


def _unique_rows(a: np.ndarray) -> np.ndarray:
    """Sorted unique rows; lexicographic row order == integer order of the codes."""
    if a.shape[0] == 0:
        return a
    return np.unique(a, axis=0)


class HipLinearHashIndex(HashIndex):
    """Linear Hamming scan over unique hash codes, executed by HIP kernels."""

    @classmethod
    def is_usable(cls) -> bool:
        return _lib.usable()

    @classmethod
    def get_default_config(cls) -> Dict[str, Any]:
        c = super().get_default_config()
        c["cache_element"] = make_default_config(DataElement.get_impls())
        return c

    @classmethod
    def from_config(cls: Type[T], config_dict: Dict, merge_default: bool = True) -> T:
        if merge_default:
            config_dict = merge_dict(cls.get_default_config(), config_dict)
        ce = config_dict.get("cache_element")
        config_dict["cache_element"] = (from_config_dict(ce, DataElement.get_impls())
                                        if ce and ce.get("type") else None)
        return super().from_config(config_dict, False)

    def __init__(self, cache_element: Optional[DataElement] = None):
        super().__init__()
        self.cache_element = cache_element
        self._codes = np.zeros((0, 1), dtype=np.uint64)   # sorted unique packed codes
        self._dev: Optional[_lib.HammingIndex] = None
        self._model_lock = threading.RLock()
        self.load_cache()

    def get_config(self) -> Dict[str, Any]:
        c = self.get_default_config()
        if self.cache_element:
            c["cache_element"] = merge_dict(c["cache_element"], to_config_dict(self.cache_element))
        return c

    # ------------------------------------------------------------------ state
    @property
    def index(self) -> Set[int]:
        """The codes as python ints (the reference's ``index`` attribute)."""
        with self._model_lock:
            return set(packed_to_ints(self._codes)) if self._codes.shape[0] else set()

    def set_codes_packed(self, codes: np.ndarray) -> None:
        """Replace the content with packed codes ``uint64[n, W]`` (deduplicated here)."""
        with self._model_lock:
            codes = np.ascontiguousarray(codes, dtype=np.uint64)
            if codes.ndim == 1:
                codes = codes[:, None]
            self._set(_unique_rows(codes))

    def _set(self, codes: np.ndarray) -> None:
        self._codes = codes
        if self._dev is not None:
            self._dev.close()
            self._dev = None

    def _device(self) -> _lib.HammingIndex:
        if self._dev is None:
            _require_usable(self)
            self._dev = _lib.HammingIndex(self._codes)
        return self._dev

    def _pack(self, hashes: Iterable[np.ndarray]) -> np.ndarray:
        rows = [np.asarray(h).astype(bool).reshape(-1) for h in hashes]
        bits = len(rows[0])
        if any(len(r) != bits for r in rows):
            raise ValueError("hash vectors of differing bit length")
        packed = pack_bits_msb(np.vstack(rows))
        w_have = self._codes.shape[1]
        if self._codes.shape[0] and packed.shape[1] != w_have:
            # widen the narrower side with leading zero words (integer value unchanged)
            w = max(w_have, packed.shape[1])
            packed = np.pad(packed, ((0, 0), (w - packed.shape[1], 0)))
        return packed

    def _align(self, packed: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        cur = self._codes
        if cur.shape[0] and cur.shape[1] < packed.shape[1]:
            cur = np.pad(cur, ((0, 0), (packed.shape[1] - cur.shape[1], 0)))
        elif not cur.shape[0]:
            cur = np.zeros((0, packed.shape[1]), dtype=np.uint64)
        return cur, packed

    # ------------------------------------------------------------------ cache
    def load_cache(self) -> None:
        with self._model_lock:
            if self.cache_element and not self.cache_element.is_empty():
                arr = np.load(BytesIO(self.cache_element.get_bytes()))
                if arr.ndim == 1:
                    if not np.issubdtype(arr.dtype, np.integer):
                        raise ValueError("hash index cache holds a non-integer array (%s)" % arr.dtype)
                    arr = ints_to_packed([int(v) for v in arr.tolist()], 1)
                self._set(_unique_rows(np.ascontiguousarray(arr, dtype=np.uint64)))

    def save_cache(self) -> None:
        with self._model_lock:
            if self.cache_element and self._codes.shape[0]:
                if self.cache_element.is_read_only():
                    raise ValueError("Cache element (%s) is read-only." % self.cache_element)
                buf = BytesIO()
                np.save(buf, self._codes)
                self.cache_element.set_bytes(buf.getvalue())

    # -------------------------------------------------------------- interface
    def count(self) -> int:
        with self._model_lock:
            return int(self._codes.shape[0])

    def _build_index(self, hashes: Iterable[np.ndarray]) -> None:
        with self._model_lock:
            packed = pack_bits_msb(np.vstack([np.asarray(h).astype(bool).reshape(-1) for h in hashes]))
            self._set(_unique_rows(packed))
            self.save_cache()

    @staticmethod
    def _locate(cur: np.ndarray, other: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """For sorted unique ``cur`` ``[n, W]`` and sorted unique ``other`` ``[m, W]``: (merged sorted unique rows,
        position of every ``other`` row in ``merged``, mask of the ``other`` rows that are in ``cur``)."""
        merged, inv = np.unique(np.vstack([cur, other]), axis=0, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
        at = inv[cur.shape[0]:]
        present = np.zeros(merged.shape[0], dtype=bool)
        present[inv[:cur.shape[0]]] = True
        return merged, at, present[at]

    def _update_index(self, hashes: Iterable[np.ndarray]) -> None:
        with self._model_lock:
            cur, new = self._align(self._pack(hashes))
            new = _unique_rows(new)
            merged, at, known = self._locate(cur, new)
            dev = self._dev
            if (dev is not None and cur.shape[0] and cur.shape[1] == self._codes.shape[1] == dev.words
                    and (~known).sum() <= cur.shape[0]):
                # only the codes that are really new travel to the device (sq_hamming_append), with the number of
                # indexed codes below each: linear.py:167-182 is a set union, not a rebuild
                fresh = np.ascontiguousarray(new[~known])
                pos = at[~known] - np.arange(fresh.shape[0], dtype=np.int64)
                self._codes = merged
                if fresh.shape[0]:
                    try:
                        dev.append(fresh, pos)
                    except _lib.HipError:
                        self._set(merged)      # (a borrowed device array cannot grow: re-upload at the next query)
            else:
                self._set(merged)
            self.save_cache()

    def _remove_from_index(self, hashes: Iterable[np.ndarray]) -> None:
        with self._model_lock:
            cur, rem = self._align(self._pack(hashes))
            rem = _unique_rows(rem)
            merged, at, known = self._locate(cur, rem)
            if not known.all():                                  # KeyError before anything changes (linear.py:197-203)
                raise KeyError(packed_to_ints(rem[~known][:1])[0])
            ranks = at.astype(np.int64)                          # every removed code is in cur: merged == cur
            keep = np.ones(cur.shape[0], dtype=bool)
            keep[ranks] = False
            left = np.ascontiguousarray(cur[keep])
            dev = self._dev
            if (dev is not None and left.shape[0] and cur.shape[1] == self._codes.shape[1] == dev.words
                    and ranks.shape[0] <= left.shape[0]):
                self._codes = left
                try:
                    dev.remove(ranks)                            # sq_hamming_remove: the ranks, not the array, cross PCIe
                except _lib.HipError:
                    self._set(left)
            else:
                self._set(left)
            self.save_cache()

    def nn_packed(self, queries: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
        """Batched search on packed queries ``uint64[nq, W]`` ->
        (differing-bit counts int32 ``[nq, k]``, row ids into the sorted code array)."""
        with self._model_lock:
            k = min(int(n), self.count())
            return self._device().search(queries, k)

    def nn_many(self, hashes: np.ndarray, n: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """Batched ``nn``: bool ``[nq, bits]`` -> (codes bool ``[nq, k, bits]``, dists ``[nq, k]``)."""
        h = np.asarray(hashes).astype(bool)
        bits = h.shape[1]
        with self._model_lock:
            if not self.count():
                raise ValueError("No index currently set to query from!")
            q = pack_bits_msb(h)
            w = self._codes.shape[1]
            if q.shape[1] < w:
                q = np.pad(q, ((0, 0), (w - q.shape[1], 0)))
            elif q.shape[1] > w:
                raise ValueError("query hash is wider than the indexed codes")
            dist, idx = self.nn_packed(q, n)
            codes = self._codes[idx.reshape(-1)]
            rows = unpack_bits_msb(codes, bits).reshape(idx.shape[0], idx.shape[1], bits)
            return rows, dist / float(bits)

    def _nn(self, h: np.ndarray, n: int = 1) -> Tuple[np.ndarray, Tuple[float, ...]]:
        rows, dists = self.nn_many(np.asarray(h).reshape(1, -1), n)
        return rows[0], tuple(float(d) for d in dists[0])

    def codes_packed(self) -> np.ndarray:
        """The sorted unique packed codes (row ids returned by ``nn_packed`` index this)."""
        return self._codes

    @staticmethod
    def words(bits: int) -> int:
        return words_for_bits(bits)

"""
``HashIndex`` plugin interface: kNN over UNIQUE hash codes (boolean bit
vectors) under normalised Hamming distance (contract:
smqtk_indexing/interfaces/hash_index.py:10-182).  Re-exports the real class
when ``smqtk_indexing`` is importable.
"""
import abc
from typing import Iterable, Sequence, Tuple

import numpy as np

from .._compat import Configurable, Pluggable
from ..utils.iter_validation import check_empty_iterable

try:  # pragma: no cover
    from smqtk_indexing.interfaces.hash_index import HashIndex  # type: ignore
except ImportError:

    class HashIndex(Configurable, Pluggable):  # type: ignore[no-redef]
        """Only unique bit vectors are indexed; ``nn`` never returns a code twice."""

        def __len__(self) -> int:
            return self.count()

        @staticmethod
        def _empty_iterable_exception() -> BaseException:
            return ValueError("No hash vectors in provided iterable.")

        def build_index(self, hashes: Iterable[np.ndarray]) -> None:
            check_empty_iterable(hashes, self._build_index,
                                 self._empty_iterable_exception())

        def update_index(self, hashes: Iterable[np.ndarray]) -> None:
            check_empty_iterable(hashes, self._update_index,
                                 self._empty_iterable_exception())

        def remove_from_index(self, hashes: Iterable[np.ndarray]) -> None:
            check_empty_iterable(hashes, self._remove_from_index,
                                 self._empty_iterable_exception())

        def nn(self, h: np.ndarray, n: int = 1) -> Tuple[np.ndarray, Sequence[float]]:
            """``n`` nearest codes as bool rows and distances in [0,1]
            (differing bits / query bit length)."""
            if not self.count():
                raise ValueError("No index currently set to query from!")
            return self._nn(h, n)

        @abc.abstractmethod
        def count(self) -> int: ...

        @abc.abstractmethod
        def _build_index(self, hashes: Iterable[np.ndarray]) -> None: ...

        @abc.abstractmethod
        def _update_index(self, hashes: Iterable[np.ndarray]) -> None: ...

        @abc.abstractmethod
        def _remove_from_index(self, hashes: Iterable[np.ndarray]) -> None: ...

        @abc.abstractmethod
        def _nn(self, h: np.ndarray, n: int = 1) -> Tuple[np.ndarray, Tuple[float, ...]]: ...

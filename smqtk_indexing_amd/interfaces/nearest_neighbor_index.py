"""
``NearestNeighborsIndex`` plugin interface.

When the real ``smqtk_indexing`` package is importable its class is re-exported
unchanged, so implementations in this package are discovered by
``NearestNeighborsIndex.get_impls()`` of a real SMQTK deployment.  Otherwise an
interface with the same template methods, messages and exception behaviour is
defined here (contract: smqtk_indexing/interfaces/nearest_neighbor_index.py:13-184).
"""
import abc
from typing import Hashable, Iterable, Tuple

from .._compat import Configurable, DescriptorElement, Pluggable
from ..utils.iter_validation import check_empty_iterable

try:  # pragma: no cover
    from smqtk_indexing.interfaces.nearest_neighbor_index import NearestNeighborsIndex  # type: ignore
except ImportError:

    class NearestNeighborsIndex(Configurable, Pluggable):  # type: ignore[no-redef]
        """Descriptor kNN over a built index.  Public methods validate and
        dispatch to the ``_``-prefixed hooks (template-method pattern);
        implementations must be thread safe."""

        def __len__(self) -> int:
            return self.count()

        @staticmethod
        def _empty_iterable_exception() -> BaseException:
            return ValueError("No DescriptorElement instances in provided "
                              "iterable.")

        def build_index(self, descriptors: Iterable[DescriptorElement]) -> None:
            """Replace the index content. ValueError on an empty iterable."""
            check_empty_iterable(descriptors, self._build_index,
                                 self._empty_iterable_exception())

        def update_index(self, descriptors: Iterable[DescriptorElement]) -> None:
            """Add to the index (build when empty). ValueError on an empty iterable."""
            check_empty_iterable(descriptors, self._update_index,
                                 self._empty_iterable_exception())

        def remove_from_index(self, uids: Iterable[Hashable]) -> None:
            """Remove by UID. ValueError on empty input; KeyError (index
            untouched) when any UID is unknown."""
            check_empty_iterable(uids, self._remove_from_index,
                                 self._empty_iterable_exception())

        def nn(self, d: DescriptorElement, n: int = 1
               ) -> Tuple[Tuple[DescriptorElement, ...], Tuple[float, ...]]:
            """``n`` nearest descriptors to ``d`` and their distances, ascending."""
            if not d.has_vector():
                raise ValueError("Query descriptor did not have a vector set!")
            elif not self.count():
                raise ValueError("No index currently set to query from!")
            return self._nn(d, n)

        @abc.abstractmethod
        def count(self) -> int:
            """Number of indexed elements."""

        @abc.abstractmethod
        def _build_index(self, descriptors: Iterable[DescriptorElement]) -> None: ...

        @abc.abstractmethod
        def _update_index(self, descriptors: Iterable[DescriptorElement]) -> None: ...

        @abc.abstractmethod
        def _remove_from_index(self, uids: Iterable[Hashable]) -> None: ...

        @abc.abstractmethod
        def _nn(self, d: DescriptorElement, n: int = 1
                ) -> Tuple[Tuple[DescriptorElement, ...], Tuple[float, ...]]: ...

"""
``LshFunctor`` plugin interface: descriptor vector -> boolean hash code
(contract: smqtk_indexing/interfaces/lsh_functor.py:11-41).  Re-exports the
real class when ``smqtk_indexing`` is importable.
"""
import abc

import numpy as np

from .._compat import Configurable, Pluggable

try:  # pragma: no cover
    from smqtk_indexing.interfaces.lsh_functor import LshFunctor  # type: ignore
except ImportError:

    class LshFunctor(Configurable, Pluggable):  # type: ignore[no-redef]
        def __call__(self, descriptor: np.ndarray) -> np.ndarray:
            return self.get_hash(descriptor)

        @abc.abstractmethod
        def get_hash(self, descriptor: np.ndarray) -> np.ndarray:
            """Bit vector (numpy bool array) for one descriptor vector."""

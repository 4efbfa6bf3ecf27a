"""
MI355X-native kNN backend behind SMQTK-Indexing's plugin surface
(NearestNeighborsIndex / HashIndex / LshFunctor).  See DESIGN.md.
"""
from .interfaces.nearest_neighbor_index import NearestNeighborsIndex  # noqa: F401
from .interfaces.lsh_functor import LshFunctor  # noqa: F401
from .interfaces.hash_index import HashIndex  # noqa: F401

__version__ = "0.1.0"

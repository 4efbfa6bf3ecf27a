"""Empty-iterable guard used by the interface template methods
(behaviour of smqtk_indexing/utils/iter_validation.py:8-28)."""
import itertools
from typing import Callable, Iterable, TypeVar

T = TypeVar("T")


def check_empty_iterable(iterable: Iterable, callback: Callable[[Iterable], T],
                         exception_inst: BaseException) -> T:
    """Peek one item; raise ``exception_inst`` when there is none, else hand
    the re-chained iterable to ``callback`` and return its result."""
    it = iter(iterable)
    for first in it:
        return callback(itertools.chain((first,), it))
    raise exception_inst

"""
Interface-compatible stand-ins for the SMQTK sibling packages the plugin
surface is written against (``smqtk_core``, ``smqtk_dataprovider``,
``smqtk_descriptors``).

The reference imports these from PyPI (poetry.lock:992-1023); they are not
part of the reference tree.  When the real packages are importable they are
used untouched, so the plugin classes of this package register with a real
SMQTK deployment.  When they are absent (as in the build/GPU containers) the
minimal classes below provide exactly the surface the hot path touches
("External types crossing the boundary"):

* ``Configurable`` / ``Pluggable`` + ``make_default_config`` /
  ``to_config_dict`` / ``from_config_dict`` / ``merge_dict``
* ``DataElement`` (+ in-memory impl), ``KeyValueStore`` (+ in-memory impl),
  ``ReadOnlyError``
* ``DescriptorElement`` (+ in-memory impl), ``DescriptorSet`` (+ in-memory
  impl), ``parallel_map``

Only behaviour, not code, follows the upstream packages.
"""
import abc
import importlib
import inspect
import os
import threading
from typing import (Any, Callable, Dict, Hashable, Iterable, Iterator, List,
                    Mapping, Optional, Sequence, Set, Type, TypeVar)

import numpy as np

try:  # pragma: no cover - exercised only where SMQTK is installed
    from smqtk_core import Configurable, Pluggable  # type: ignore
    from smqtk_core.configuration import (  # type: ignore
        from_config_dict, make_default_config, to_config_dict,
    )
    from smqtk_core.dict import merge_dict  # type: ignore
    from smqtk_dataprovider import DataElement, KeyValueStore  # type: ignore
    from smqtk_dataprovider.exceptions import ReadOnlyError  # type: ignore
    from smqtk_dataprovider.impls.data_element.memory import (  # type: ignore
        DataMemoryElement,
    )
    from smqtk_dataprovider.impls.key_value_store.memory import (  # type: ignore
        MemoryKeyValueStore,
    )
    from smqtk_descriptors import DescriptorElement, DescriptorSet  # type: ignore
    from smqtk_descriptors.impls.descriptor_element.memory import (  # type: ignore
        DescriptorMemoryElement,
    )
    from smqtk_descriptors.impls.descriptor_set.memory import (  # type: ignore
        MemoryDescriptorSet,
    )
    from smqtk_descriptors.utils import parallel_map  # type: ignore
    HAVE_SMQTK = True
except ImportError:
    HAVE_SMQTK = False

T = TypeVar("T")

if not HAVE_SMQTK:

    # ------------------------------------------------------------------ dict
    def merge_dict(a: Dict, b: Mapping, deep_copy: bool = False) -> Dict:
        """Recursively overlay ``b`` onto ``a`` (in place) and return ``a``."""
        import copy
        for key, val in b.items():
            if isinstance(a.get(key), dict) and isinstance(val, Mapping):
                merge_dict(a[key], val, deep_copy)
            else:
                a[key] = copy.deepcopy(val) if deep_copy else val
        return a

    # ---------------------------------------------------------- configurable
    class Configurable(metaclass=abc.ABCMeta):
        """JSON-dictionary <-> constructor round trip."""

        @classmethod
        def get_default_config(cls) -> Dict[str, Any]:
            sig = inspect.signature(cls.__init__)
            out: Dict[str, Any] = {}
            for name, p in list(sig.parameters.items())[1:]:
                if p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
                    continue
                out[name] = None if p.default is p.empty else p.default
            return out

        @classmethod
        def from_config(cls: Type[T], config_dict: Dict,
                        merge_default: bool = True) -> T:
            if merge_default:
                config_dict = merge_dict(cls.get_default_config(), config_dict)
            return cls(**config_dict)  # type: ignore

        @abc.abstractmethod
        def get_config(self) -> Dict[str, Any]:
            """JSON-compliant dictionary that reconstructs this instance."""

    # -------------------------------------------------------------- pluggable
    def _all_subclasses(cls: type) -> Set[type]:
        found: Set[type] = set()
        stack = list(cls.__subclasses__())
        while stack:
            c = stack.pop()
            if c not in found:
                found.add(c)
                stack.extend(c.__subclasses__())
        return found

    class Pluggable(metaclass=abc.ABCMeta):
        """Implementation discovery: env ``SMQTK_PLUGIN_PATH`` modules, the
        ``smqtk_plugins`` entry-point group, then every loaded subclass."""

        PLUGIN_ENV_VAR = "SMQTK_PLUGIN_PATH"
        PLUGIN_NAMESPACE = "smqtk_plugins"

        @classmethod
        def is_usable(cls) -> bool:
            return True

        @classmethod
        def get_impls(cls: Type[T]) -> Set[Type[T]]:
            for mod in filter(None, os.environ.get(
                    Pluggable.PLUGIN_ENV_VAR, "").split(os.pathsep)):
                try:
                    importlib.import_module(mod)
                except ImportError:
                    pass
            try:
                from importlib import metadata
                eps = metadata.entry_points()
                group = (eps.select(group=Pluggable.PLUGIN_NAMESPACE)
                         if hasattr(eps, "select")
                         else eps.get(Pluggable.PLUGIN_NAMESPACE, []))
                for ep in group:
                    try:
                        ep.load()
                    except Exception:
                        pass
            except Exception:
                pass
            return {c for c in _all_subclasses(cls)
                    if not inspect.isabstract(c) and c.is_usable()}

    # ---------------------------------------------------------- configuration
    def _type_key(cls: type) -> str:
        return f"{cls.__module__}.{cls.__name__}"

    def make_default_config(configurable_iter: Iterable[type]) -> Dict[str, Any]:
        d: Dict[str, Any] = {"type": None}
        for c in configurable_iter:
            d[_type_key(c)] = c.get_default_config()
        return d

    def to_config_dict(c_inst: Any) -> Dict[str, Any]:
        key = _type_key(type(c_inst))
        return {"type": key, key: c_inst.get_config()}

    def from_config_dict(config: Dict, type_iter: Iterable[type], *args: Any) -> Any:
        if "type" not in config:
            raise ValueError("Configuration block has no 'type' key.")
        t = config["type"]
        if t is None:
            raise ValueError("No implementation type selected ('type' is None).")
        if t not in config:
            raise ValueError(f"Type '{t}' has no configuration block.")
        by_key = {_type_key(c): c for c in type_iter}
        if t not in by_key:
            raise ValueError(f"Type '{t}' is not an available implementation.")
        return by_key[t].from_config(config[t], *args)

    # ------------------------------------------------------------ exceptions
    class ReadOnlyError(Exception):
        """Mutation attempted on a read-only container."""

    # ------------------------------------------------------------ data element
    class DataElement(Configurable, Pluggable):
        """Byte container abstraction (only the methods the path uses)."""

        @abc.abstractmethod
        def is_empty(self) -> bool: ...

        @abc.abstractmethod
        def get_bytes(self) -> bytes: ...

        @abc.abstractmethod
        def set_bytes(self, b: bytes) -> None: ...

        @abc.abstractmethod
        def writable(self) -> bool: ...

        def is_read_only(self) -> bool:
            return not self.writable()

    class DataMemoryElement(DataElement):
        def __init__(self, bytes: Optional[bytes] = None,  # noqa: A002
                     content_type: Optional[str] = None, readonly: bool = False):
            self._bytes = bytes
            self._content_type = content_type
            self._readonly = bool(readonly)

        def get_config(self) -> Dict[str, Any]:
            import base64
            b = self._bytes
            return {
                "bytes": base64.b64encode(b).decode() if b is not None else None,
                "content_type": self._content_type,
                "readonly": self._readonly,
            }

        @classmethod
        def from_config(cls, config_dict: Dict, merge_default: bool = True) -> "DataMemoryElement":
            import base64
            c = dict(config_dict)
            if isinstance(c.get("bytes"), str):
                c["bytes"] = base64.b64decode(c["bytes"])
            return super().from_config(c, merge_default)

        def is_empty(self) -> bool:
            return not self._bytes

        def get_bytes(self) -> bytes:
            return self._bytes or b""

        def set_bytes(self, b: bytes) -> None:
            if self._readonly:
                raise ReadOnlyError("This memory element cannot be written to.")
            self._bytes = b

        def writable(self) -> bool:
            return not self._readonly

    # --------------------------------------------------------- key-value store
    _NO_DEFAULT = object()

    class KeyValueStore(Configurable, Pluggable):
        NO_DEFAULT_VALUE = _NO_DEFAULT

        def __len__(self) -> int:
            return self.count()

        def __contains__(self, k: Hashable) -> bool:
            return self.has(k)

        def __getitem__(self, k: Hashable) -> Any:
            return self.get(k)

        @abc.abstractmethod
        def count(self) -> int: ...

        @abc.abstractmethod
        def keys(self) -> Iterator[Hashable]: ...

        def values(self) -> Iterator[Any]:
            for k in self.keys():
                yield self.get(k)

        @abc.abstractmethod
        def is_read_only(self) -> bool: ...

        @abc.abstractmethod
        def has(self, key: Hashable) -> bool: ...

        def _guard(self) -> None:
            if self.is_read_only():
                raise ReadOnlyError("Cannot modify a read-only key-value store.")

        def add(self, key: Hashable, value: Any) -> "KeyValueStore":
            return self.add_many({key: value})

        @abc.abstractmethod
        def add_many(self, d: Mapping[Hashable, Any]) -> "KeyValueStore": ...

        def remove(self, key: Hashable) -> "KeyValueStore":
            return self.remove_many([key])

        @abc.abstractmethod
        def remove_many(self, keys: Iterable[Hashable]) -> "KeyValueStore": ...

        @abc.abstractmethod
        def get(self, key: Hashable, default: Any = _NO_DEFAULT) -> Any: ...

        def get_many(self, keys: Iterable[Hashable],
                     default: Any = _NO_DEFAULT) -> Iterator[Any]:
            for k in keys:
                yield self.get(k, default)

        @abc.abstractmethod
        def clear(self) -> "KeyValueStore": ...

    class MemoryKeyValueStore(KeyValueStore):
        def __init__(self, cache_element: Optional[DataElement] = None):
            self._table: Dict[Hashable, Any] = {}
            self._cache_element = cache_element
            self._lock = threading.RLock()

        def get_config(self) -> Dict[str, Any]:
            return {"cache_element": None}

        def count(self) -> int:
            return len(self._table)

        def keys(self) -> Iterator[Hashable]:
            return iter(list(self._table.keys()))

        def values(self) -> Iterator[Any]:
            return iter(list(self._table.values()))

        def is_read_only(self) -> bool:
            return False

        def has(self, key: Hashable) -> bool:
            return key in self._table

        def add_many(self, d: Mapping[Hashable, Any]) -> "MemoryKeyValueStore":
            self._guard()
            with self._lock:
                self._table.update(d)
            return self

        def remove_many(self, keys: Iterable[Hashable]) -> "MemoryKeyValueStore":
            self._guard()
            keys = list(keys)
            with self._lock:
                missing = [k for k in keys if k not in self._table]
                if missing:
                    raise KeyError(missing)
                for k in keys:
                    del self._table[k]
            return self

        def get(self, key: Hashable, default: Any = _NO_DEFAULT) -> Any:
            with self._lock:
                if key in self._table:
                    return self._table[key]
            if default is _NO_DEFAULT:
                raise KeyError(key)
            return default

        def clear(self) -> "MemoryKeyValueStore":
            self._guard()
            with self._lock:
                self._table.clear()
            return self

    # ------------------------------------------------------ descriptor element
    class DescriptorElement(Configurable, Pluggable):
        def __init__(self, uuid: Hashable):
            self._uuid = uuid

        def uuid(self) -> Hashable:
            return self._uuid

        def __hash__(self) -> int:
            return hash(self._uuid)

        def __eq__(self, other: Any) -> bool:
            if not isinstance(other, DescriptorElement):
                return False
            a, b = self.vector(), other.vector()
            if a is None or b is None:
                return a is None and b is None and self.uuid() == other.uuid()
            return self.uuid() == other.uuid() and np.array_equal(a, b)

        def __ne__(self, other: Any) -> bool:
            return not (self == other)

        @abc.abstractmethod
        def has_vector(self) -> bool: ...

        @abc.abstractmethod
        def vector(self) -> Optional[np.ndarray]: ...

        @abc.abstractmethod
        def set_vector(self, new_vec: np.ndarray) -> "DescriptorElement": ...

        @classmethod
        def get_many_vectors(cls, descriptors: Iterable["DescriptorElement"]
                             ) -> List[Optional[np.ndarray]]:
            return [d.vector() for d in descriptors]

    class DescriptorMemoryElement(DescriptorElement):
        def __init__(self, uuid: Hashable):
            super().__init__(uuid)
            self._v: Optional[np.ndarray] = None

        def get_config(self) -> Dict[str, Any]:
            return {}

        def has_vector(self) -> bool:
            return self._v is not None

        def vector(self) -> Optional[np.ndarray]:
            return self._v

        def set_vector(self, new_vec: np.ndarray) -> "DescriptorMemoryElement":
            self._v = None if new_vec is None else np.array(new_vec)
            return self

        def __repr__(self) -> str:
            return f"DescriptorMemoryElement{{uuid: {self._uuid}}}"

    # ---------------------------------------------------------- descriptor set
    class DescriptorSet(Configurable, Pluggable):
        def __len__(self) -> int:
            return self.count()

        def __contains__(self, item: Any) -> bool:
            if isinstance(item, DescriptorElement):
                return self.has_descriptor(item.uuid())
            return False

        def __iter__(self) -> Iterator[DescriptorElement]:
            return self.iterdescriptors()

        def __getitem__(self, uuid: Hashable) -> DescriptorElement:
            return self.get_descriptor(uuid)

        def get_many_vectors(self, uuids: Iterable[Hashable]) -> List[Optional[np.ndarray]]:
            return DescriptorElement.get_many_vectors(self.get_many_descriptors(uuids))

        @abc.abstractmethod
        def count(self) -> int: ...

        @abc.abstractmethod
        def clear(self) -> None: ...

        @abc.abstractmethod
        def has_descriptor(self, uuid: Hashable) -> bool: ...

        @abc.abstractmethod
        def add_descriptor(self, descriptor: DescriptorElement) -> None: ...

        @abc.abstractmethod
        def add_many_descriptors(self, descriptors: Iterable[DescriptorElement]) -> None: ...

        @abc.abstractmethod
        def get_descriptor(self, uuid: Hashable) -> DescriptorElement: ...

        @abc.abstractmethod
        def get_many_descriptors(self, uuids: Iterable[Hashable]) -> Iterator[DescriptorElement]: ...

        @abc.abstractmethod
        def remove_descriptor(self, uuid: Hashable) -> None: ...

        @abc.abstractmethod
        def remove_many_descriptors(self, uuids: Iterable[Hashable]) -> None: ...

        @abc.abstractmethod
        def keys(self) -> Iterator[Hashable]: ...

        @abc.abstractmethod
        def iterdescriptors(self) -> Iterator[DescriptorElement]: ...

        def items(self) -> Iterator:
            for d in self.iterdescriptors():
                yield d.uuid(), d

    class MemoryDescriptorSet(DescriptorSet):
        def __init__(self, cache_element: Optional[DataElement] = None,
                     pickle_protocol: int = -1):
            self._table: Dict[Hashable, DescriptorElement] = {}
            self.cache_element = cache_element
            self.pickle_protocol = pickle_protocol

        def get_config(self) -> Dict[str, Any]:
            return {"cache_element": None, "pickle_protocol": self.pickle_protocol}

        def count(self) -> int:
            return len(self._table)

        def clear(self) -> None:
            self._table = {}

        def has_descriptor(self, uuid: Hashable) -> bool:
            return uuid in self._table

        def add_descriptor(self, descriptor: DescriptorElement) -> None:
            self._table[descriptor.uuid()] = descriptor

        def add_many_descriptors(self, descriptors: Iterable[DescriptorElement]) -> None:
            added = {d.uuid(): d for d in descriptors}
            self._table.update(added)

        def get_descriptor(self, uuid: Hashable) -> DescriptorElement:
            return self._table[uuid]

        def get_many_descriptors(self, uuids: Iterable[Hashable]) -> Iterator[DescriptorElement]:
            # KeyError surfaces before anything is yielded, like a dict lookup.
            found = [self._table[u] for u in uuids]
            return iter(found)

        def remove_descriptor(self, uuid: Hashable) -> None:
            del self._table[uuid]

        def remove_many_descriptors(self, uuids: Iterable[Hashable]) -> None:
            uuids = list(uuids)
            for u in uuids:
                if u not in self._table:
                    raise KeyError(u)
            for u in uuids:
                del self._table[u]

        def keys(self) -> Iterator[Hashable]:
            return iter(list(self._table.keys()))

        def iterdescriptors(self) -> Iterator[DescriptorElement]:
            return iter(list(self._table.values()))

    # ------------------------------------------------------------------ utils
    def parallel_map(work_func: Callable, *sequences: Iterable, **_kw: Any) -> Iterator:
        """Ordered map; the thread/process pool of the upstream helper only
        parallelises fetching vectors, which is irrelevant for in-memory
        elements."""
        return map(work_func, *sequences)


__all__ = [
    "HAVE_SMQTK", "Configurable", "Pluggable", "merge_dict",
    "make_default_config", "to_config_dict", "from_config_dict",
    "ReadOnlyError", "DataElement", "DataMemoryElement", "KeyValueStore",
    "MemoryKeyValueStore", "DescriptorElement", "DescriptorMemoryElement",
    "DescriptorSet", "MemoryDescriptorSet", "parallel_map",
]

"""
ITQ locality-sensitive hash functor whose ``get_hash`` runs on MI355X.

Drop-in counterpart of ``ItqFunctor``
(smqtk_indexing/impls/lsh_functor/itq.py:31-408): same constructor
arguments, same model files (``mean_vec`` / ``rotation`` as ``.npy`` bytes in
``DataElement`` caches, itq.py:212-237), same bit convention (bit 0 = most
significant, itq.py:46-50) and the same sign rule (``z >= 0`` -> True,
itq.py:407).  ``get_hash`` calls ``sq_itq_hash`` (include/smqtk_hip.h); there
is no CPU code path for hashing.

``fit`` (itq.py:291-387) is model *training*, outside the query/build hot path
(SURVEY.md section 8f rank 3): it runs on the host with numpy exactly as the
reference does, then the codes of the training set are produced on the GPU.
"""
from io import BytesIO
import logging
from typing import Any, Dict, Iterable, Optional, Tuple, Type, TypeVar, Union

import numpy as np

from .. import _require_usable
from ... import _lib
from ..._compat import (DataElement, DescriptorElement, from_config_dict,
                        make_default_config, merge_dict, to_config_dict)
from ...interfaces.lsh_functor import LshFunctor
from ...utils.bits import unpack_bits_msb

LOG = logging.getLogger(__name__)
T = TypeVar("T", bound="HipItqFunctor")


class HipItqFunctor(LshFunctor):
    """ITQ (Gong & Lazebnik, CVPR 2011) hash codes; rotation + sign on the GPU."""

    @classmethod
    def is_usable(cls) -> bool:
        return _lib.usable()

    @classmethod
    def get_default_config(cls) -> Dict[str, Any]:
        c = super().get_default_config()
        slots = make_default_config(DataElement.get_impls())
        c["mean_vec_cache"] = slots
        c["rotation_cache"] = dict(slots)
        return c

    @classmethod
    def from_config(cls: Type[T], config_dict: Dict, merge_default: bool = True) -> T:
        if merge_default:
            config_dict = merge_dict(cls.get_default_config(), config_dict)
        for slot in ("mean_vec_cache", "rotation_cache"):
            v = config_dict.get(slot)
            config_dict[slot] = (from_config_dict(v, DataElement.get_impls())
                                 if v and v.get("type") else None)
        return super().from_config(config_dict, False)

    def __init__(self, mean_vec_cache: Optional[DataElement] = None,
                 rotation_cache: Optional[DataElement] = None,
                 bit_length: int = 8, itq_iterations: int = 50,
                 normalize: Optional[Union[int, float, str]] = None,
                 random_seed: Optional[int] = None, fit_on_device: bool = True):
        super().__init__()
        self.fit_on_device = bool(fit_on_device)
        self.mean_vec_cache_elem = mean_vec_cache
        self.rotation_cache_elem = rotation_cache
        self.bit_length = bit_length
        self.itq_iterations = itq_iterations
        self.normalize = normalize
        self.random_seed = random_seed
        if normalize is not None:
            # the validation the reference performs (itq.py:162-164): whatever numpy.linalg.norm accepts for a vector
            self._norm_vector(np.random.rand(8))
        self.mean_vec: Optional[np.ndarray] = None
        self.rotation: Optional[np.ndarray] = None
        self.load_model()

    # ------------------------------------------------------------------ model
    def get_config(self) -> Dict[str, Any]:
        c = merge_dict(self.get_default_config(), {
            "bit_length": self.bit_length,
            "itq_iterations": self.itq_iterations,
            "normalize": self.normalize,
            "random_seed": self.random_seed,
            "fit_on_device": self.fit_on_device,
        })
        if self.mean_vec_cache_elem:
            c["mean_vec_cache"] = to_config_dict(self.mean_vec_cache_elem)
        if self.rotation_cache_elem:
            c["rotation_cache"] = to_config_dict(self.rotation_cache_elem)
        return c

    def has_model(self) -> bool:
        return self.mean_vec is not None and self.rotation is not None

    def load_model(self) -> None:
        m, r = self.mean_vec_cache_elem, self.rotation_cache_elem
        if m and r and not m.is_empty() and not r.is_empty():
            self.mean_vec = np.load(BytesIO(m.get_bytes()))
            self.rotation = np.load(BytesIO(r.get_bytes()))

    def save_model(self) -> None:
        m, r = self.mean_vec_cache_elem, self.rotation_cache_elem
        if m and r and m.writable() and r.writable() and self.has_model():
            for elem, arr in ((m, self.mean_vec), (r, self.rotation)):
                buf = BytesIO()
                np.save(buf, arr)
                elem.set_bytes(buf.getvalue())

    # normalize values whose row norm the device evaluates in numpy's own arithmetic (include/smqtk_hip.h); any other
    # order numpy accepts (a general p: |x|**p through libm's pow) is normalised here by numpy itself, exactly as the
    # reference does (itq.py:185-188), and the normalised rows are hashed with SQ_NORM_NONE
    _DEVICE_NORMS = {2.0: _lib.SQ_NORM_L2, 1.0: _lib.SQ_NORM_L1, 0.0: _lib.SQ_NORM_L0,
                     float("inf"): _lib.SQ_NORM_INF, float("-inf"): _lib.SQ_NORM_NEG_INF}

    def _norm_on_host(self) -> bool:
        return self.normalize is not None and self._norm_ord() == _lib.SQ_NORM_NONE

    def _norm_ord(self) -> int:
        if self.normalize is None:
            return _lib.SQ_NORM_NONE
        try:
            return self._DEVICE_NORMS.get(float(self.normalize), _lib.SQ_NORM_NONE)
        except (TypeError, ValueError):
            return _lib.SQ_NORM_NONE

    def _norm_vector(self, v: np.ndarray) -> np.ndarray:
        """Host normalisation used by ``fit`` only (itq.py:172-191)."""
        if self.normalize is None:
            return v
        n = np.linalg.norm(v, self.normalize, v.ndim - 1, keepdims=True)
        n[n == 0.] = 1.
        return v / n

    # ---------------------------------------------------------------- hashing
    def get_hash_packed(self, descriptors: np.ndarray) -> np.ndarray:
        """Packed codes ``uint64[n, ceil(bits/64)]`` of an ``[n, d]`` matrix (GPU)."""
        _require_usable(self)
        if not self.has_model():
            if self.mean_vec is None:
                raise Exception("Can't compute hash code: mean vector is none.")
            raise Exception("Can't compute hash code: rotation matrix is none.")
        x = np.asarray(descriptors)
        if x.ndim != 2:
            raise ValueError("expected an [n, d] matrix")
        if x.dtype != np.float32:
            x = x.astype(np.float64)       # ints / float16 etc. upcast like numpy would
        if self._norm_on_host():
            x = np.ascontiguousarray(self._norm_vector(x), dtype=x.dtype)
        return self._device_model().hash(x)

    def _device_model(self) -> "_lib.ItqModel":
        """The model resident on the device (sq_itq_model_*), rebuilt when mean_vec / rotation / normalize change
        (they are plain attributes, as in the reference, so identity and norm are what can be checked)."""
        # the model arrays are plain attributes, as in the reference: an in-place edit keeps their identity, so the
        # key carries a digest of their content as well (d * bits * 8 bytes: microseconds)
        import hashlib
        digest = hashlib.blake2b(np.ascontiguousarray(self.mean_vec).tobytes() +
                                 np.ascontiguousarray(np.real(self.rotation)).tobytes(), digest_size=16).digest()
        key = (id(self.mean_vec), id(self.rotation), self._norm_ord(), digest)
        cached = getattr(self, "_model_cache", None)
        if cached is None or cached[0] != key:
            if cached is not None:
                cached[1].close()
            # the arrays are kept alive with the key so that an id cannot be reused by another array
            cached = (key, _lib.ItqModel(self.mean_vec, np.real(self.rotation), self._norm_ord()), self.mean_vec, self.rotation)
            self._model_cache = cached
        return cached[1]

    def get_hash(self, descriptor: np.ndarray) -> np.ndarray:
        """Boolean hash vector of one descriptor (or bool ``[n, bits]`` of a matrix)."""
        x = np.asarray(descriptor)
        single = x.ndim == 1
        if self.mean_vec is None:
            raise Exception("Can't compute hash code: mean vector is none.")
        elif self.rotation is None:
            raise Exception("Can't compute hash code: rotation matrix is none.")
        bits = self.rotation.shape[1]
        b = unpack_bits_msb(self.get_hash_packed(x[None, :] if single else x), bits)
        return b[0] if single else b

    # ------------------------------------------------------------------- fit
    def _find_itq_rotation(self, v: np.ndarray, n_iter: int) -> np.ndarray:
        """Orthogonal Procrustes iterations of ITQ on the PCA-embedded data
        (itq.py:239-289): random orthogonal start (SVD of a seeded Gaussian),
        then alternate B = sign(V R) and R = argmin ||B - V R||_F."""
        nbits = v.shape[1]
        if self.random_seed is not None:
            np.random.seed(self.random_seed)
        u, _, _ = np.linalg.svd(np.random.randn(nbits, nbits))
        r = u[:, :nbits]
        for _ in range(n_iter):
            b = np.where(np.dot(v, r) >= 0, 1.0, -1.0)
            ub, _, ua = np.linalg.svd(np.dot(b.T, v))
            r = np.dot(ua, ub.T)
        return r

    def _fit_device(self, x_in: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """(mean_vec, rotation) with the O(n) work of itq.py:330-362 and 271-275 on the device
        (``sq_itqfit_*``): mean, covariance, PCA projection and, per ITQ iteration, sign(V R) and
        B^T V.  The d x d eigen-decomposition and the bits x bits SVDs are numpy's, as in the
        reference.  Descriptors up to 512-d, codes up to 256 bits."""
        nbits = self.bit_length
        norm_ord = self._norm_ord()
        if norm_ord not in (_lib.SQ_NORM_NONE, _lib.SQ_NORM_L2) or self._norm_on_host():
            # the training products know normalize=None / 2; every other order is applied by numpy first
            x_in = np.ascontiguousarray(self._norm_vector(x_in))
            norm_ord = _lib.SQ_NORM_NONE
        fit = _lib.ItqFit(x_in, norm_ord)
        try:
            # np.mean keeps the descriptors' dtype; the model (and everything derived below) uses that value
            mean_vec = fit.mean.astype(x_in.dtype if x_in.dtype in (np.float32, np.float64) else np.float64)
            fit.set_mean(mean_vec)
            evals, evecs = np.linalg.eig(np.atleast_2d(fit.cov()))
            ranked = sorted(zip(evals, evecs.T), key=lambda p: p[0], reverse=True)
            pc_top = np.array([p[1] for p in ranked[:nbits]]).T
            fit.project(np.real(pc_top))
            if self.random_seed is not None:
                np.random.seed(self.random_seed)
            u, _, _ = np.linalg.svd(np.random.randn(nbits, nbits))
            r = u[:, :nbits]
            for _ in range(self.itq_iterations):
                ub, _, ua = np.linalg.svd(fit.iterate(r))
                r = np.dot(ua, ub.T)
        finally:
            fit.close()
        return mean_vec, np.dot(pc_top, r)

    def fit(self, descriptors: Iterable[DescriptorElement], use_multiprocessing: bool = True) -> np.ndarray:
        """Train mean vector and rotation from descriptors, then return the training set's codes
        (bool ``[n, bits]``, computed on the GPU).  With ``fit_on_device`` (default) the products
        over the n descriptors run on the device when they fit its kernels (d <= 512, <= 256 bits,
        a real-valued PCA basis); otherwise, and always for the small dense linear algebra, numpy."""
        if self.has_model():
            raise RuntimeError("Model components have already been loaded.")
        descr = descriptors if isinstance(descriptors, (list, tuple)) else list(descriptors)
        if len(descr[0].vector()) < self.bit_length:
            raise ValueError("Input descriptors have fewer features than "
                             "requested bit encoding. Hash codes will be "
                             "smaller than requested due to PCA decomposition "
                             "result being bound by number of features.")
        x_in = np.asarray([d.vector() for d in descr])
        if (self.fit_on_device and x_in.ndim == 2 and x_in.shape[1] <= 512 and self.bit_length <= 256
                and x_in.shape[0] > 1 and _lib.usable()):
            self.mean_vec, self.rotation = self._fit_device(x_in)
        else:
            x = self._norm_vector(x_in)
            mean_vec = np.mean(x, axis=0)
            x = x - mean_vec
            cov = np.atleast_2d(np.cov(x.T))
            evals, evecs = np.linalg.eig(cov)
            ranked = sorted(zip(evals, evecs.T), key=lambda p: p[0], reverse=True)
            pc_top = np.array([p[1] for p in ranked[:self.bit_length]]).T
            r = self._find_itq_rotation(np.dot(x, pc_top), self.itq_iterations)
            self.mean_vec = mean_vec
            self.rotation = np.dot(pc_top, r)
        self.save_model()
        return self.get_hash(x_in)

"""
Hash-code packing shared by the host classes and the HIP kernels.

Convention (smqtk_indexing/utils/bits.py:4-56, impls/lsh_functor/itq.py:46-50):
element 0 of a bit vector is the MOST significant bit of the integer key.  The
device format is ``uint64[n, W]`` with ``W = ceil(bits/64)``, word 0 most
significant, the code right-aligned (zero padding in the top of word 0), so the
python-int key of a row is ``sum(word[i] << 64*(W-1-i))`` and lexicographic
order of rows equals integer order of keys.
"""
from typing import Iterable, List

import numpy as np


def words_for_bits(bits: int) -> int:
    return (int(bits) + 63) // 64


def pack_bits_msb(bitvecs: np.ndarray) -> np.ndarray:
    """bool/0-1 array [n,b] (or [b]) -> uint64 [n,W]."""
    a = np.asarray(bitvecs)
    if a.ndim == 1:
        a = a[None, :]
    a = a.astype(bool)
    n, b = a.shape
    w = words_for_bits(b)
    wide = np.zeros((n, w * 64), dtype=np.uint8)
    wide[:, w * 64 - b:] = a
    return (np.packbits(wide, axis=1, bitorder="big")
            .reshape(n, w, 8).view(">u8").reshape(n, w).astype(np.uint64))


def unpack_bits_msb(words: np.ndarray, bits: int) -> np.ndarray:
    """uint64 [n,W] -> bool [n,bits]."""
    wv = np.asarray(words, dtype=np.uint64)
    if wv.ndim == 1:
        wv = wv[None, :]
    n, w = wv.shape
    full = np.unpackbits(wv.astype(">u8").view(np.uint8).reshape(n, w * 8), axis=1, bitorder="big")
    return full[:, w * 64 - bits:].astype(bool)


def packed_to_ints(words: np.ndarray) -> List[int]:
    """uint64 [n,W] -> list of python ints (the hash2uuids key type)."""
    wv = np.asarray(words, dtype=np.uint64)
    if wv.ndim == 1:
        wv = wv[None, :]
    out = [0] * wv.shape[0]
    for col in wv.T.tolist():
        out = [(o << 64) | c for o, c in zip(out, col)]
    return out


def ints_to_packed(values: Iterable[int], words: int) -> np.ndarray:
    vals = list(values)
    out = np.zeros((len(vals), words), dtype=np.uint64)
    mask = (1 << 64) - 1
    for r, v in enumerate(vals):
        v = int(v)
        for c in range(words - 1, -1, -1):
            out[r, c] = v & mask
            v >>= 64
        if v:
            raise ValueError("integer key wider than %d bits" % (64 * words))
    return out


def bit_vector_to_int_large(v: np.ndarray) -> int:
    """bool[b] -> python int (MSB first)."""
    return packed_to_ints(pack_bits_msb(np.asarray(v)))[0]


def int_to_bit_vector_large(integer: int, bits: int = 0) -> np.ndarray:
    """python int -> bool[bits or minimal] (MSB first); ValueError if it does not fit."""
    size = max(int(integer).bit_length(), 1)
    if bits and bits < size:
        raise ValueError("%d bits too small to represent integer value %d." % (bits, integer))
    n = bits or size
    return unpack_bits_msb(ints_to_packed([integer], words_for_bits(n)), n)[0]

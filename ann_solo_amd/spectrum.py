"""Feature-hash encoder -- host-side mirror of the reference's
``ann_solo/spectrum.py`` (``get_dim`` :122-143, ``hash_idx`` :146-163,
``spectrum_to_vector`` :166-214) on top of the HIP kernel ``asl_encode_batch``.
Same names, argument meaning and error behaviour; the arithmetic runs on the GPU.
"""
import ctypes as C
import functools

import numpy as np

from . import _lib

HASH_SEED = 42


@functools.lru_cache(maxsize=None)
def get_dim(min_mz, max_mz, bin_size):
    """(number of bins, inclusive lower bound, exclusive upper bound) of the m/z grid."""
    n, s, e = C.c_int64(), C.c_double(), C.c_double()
    _lib.check(_lib.lib().asl_get_dim(float(min_mz), float(max_mz), float(bin_size),
                                      C.byref(n), C.byref(s), C.byref(e)))
    return n.value, s.value, e.value


@functools.lru_cache(maxsize=None)
def hash_idx(bin_idx: int, hash_len: int) -> int:
    """murmur3_32(str(bin_idx), seed 42) % hash_len."""
    return _lib.lib().asl_hash_idx(int(bin_idx), int(hash_len), HASH_SEED)


def spectra_to_vectors(mz, intensity, offsets, min_mz, max_mz, bin_size, hash_len,
                       norm=True, out=None):
    """Batch encoder. ``mz``/``intensity``/``offsets`` are numpy arrays or torch tensors
    (host or device); returns/fills ``out`` [n, hash_len] float32 (numpy unless ``out`` given)."""
    if hash_len is None:
        raise ValueError('hash_len=None (un-hashed vectors) is not supported on the device path')
    _, min_bound, _ = get_dim(min_mz, max_mz, bin_size)
    n = len(offsets) - 1
    if out is None:
        out = np.empty((n, hash_len), np.float32)
    if tuple(out.shape) != (n, hash_len):
        raise ValueError('Incorrect vector dimensionality')
    _lib.check(_lib.lib().asl_encode_batch(_lib.ptr(mz), _lib.ptr(intensity), _lib.ptr(offsets),
                                           n, min_bound, float(bin_size), int(hash_len),
                                           HASH_SEED, int(bool(norm)), _lib.ptr(out)))
    return out


def spectrum_to_vector(spectrum, min_mz: float, max_mz: float, bin_size: float,
                       hash_len: int, norm: bool = True, vector: np.ndarray = None) -> np.ndarray:
    """Drop-in for the reference function: one spectrum -> hashed float32 vector
    (written into ``vector`` if given, like the reference does with a matrix row)."""
    mz = np.ascontiguousarray(spectrum.mz, np.float32)
    inten = np.ascontiguousarray(spectrum.intensity, np.float32)
    if hash_len is None:
        raise ValueError('hash_len=None (un-hashed vectors) is not supported on the device path')
    if vector is not None and vector.shape[0] != hash_len:
        raise ValueError('Incorrect vector dimensionality')
    out = spectra_to_vectors(mz, inten, np.array([0, len(mz)], np.int32), min_mz, max_mz,
                             bin_size, hash_len, norm)[0]
    if vector is not None:
        # the reference accumulates into the caller's (zeroed) row
        vector[:] = out if not np.any(vector) else _accumulate(vector, spectrum, min_mz, max_mz,
                                                                bin_size, hash_len, norm)
        return vector
    return out


def _accumulate(vector, spectrum, min_mz, max_mz, bin_size, hash_len, norm):
    """Rare path: caller passed a non-zero ``vector``; the reference adds into it and
    then normalises the sum. Encode un-normalised on the device, add, normalise."""
    raw = spectrum_to_vector(spectrum, min_mz, max_mz, bin_size, hash_len, False)
    acc = vector + raw
    if norm:
        acc = acc / np.linalg.norm(acc)
    return acc

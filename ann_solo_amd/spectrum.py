"""Feature-hash encoder -- host-side mirror of the reference's
``ann_solo/spectrum.py`` (``get_dim`` :122-143, ``hash_idx`` :146-163,
``spectrum_to_vector`` :166-214) on top of the HIP kernel ``asl_encode_batch``.
Same names, argument meaning and error behaviour; the arithmetic runs on the GPU.
"""
import ctypes as C
import functools
from dataclasses import dataclass
from typing import Any, List, Optional

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


def process_spectra(raw, is_library: bool, config=None, device='cuda'):
    """Batched ``process_spectrum`` (reference spectrum.py:57-119) on the device.

    ``raw``: PackedSpectra of raw peaks (ascending m/z); ``config``: an object with the
    reference's preprocessing flags (``ann_solo_amd.spectral_library.Config`` lacks them, so
    the reference defaults are used for missing attributes). Returns ``(processed
    PackedSpectra on ``device``, valid bool tensor)``; invalid spectra come back empty.
    Peak charge annotations follow their peaks."""
    import ctypes as C
    import torch
    from .packed import PackedSpectra
    from .config import Config
    config = Config.from_reference(config)
    g = lambda k, dflt: getattr(config, k, dflt)
    scaling = g('scaling', 'rank')
    scaling = {'rank': 1, 'sqrt': 2, 'root': 2, None: 0}[scaling]
    max_peaks = g('max_peaks_used_library', 50) if is_library else g('max_peaks_used', 50)
    resolution = g('resolution', None)
    P = _lib.AslProcessParams(float(g('min_mz', 11)), float(g('max_mz', 2010)),
                              int(bool(g('remove_precursor', False))),
                              float(g('remove_precursor_tolerance', 0.0)),
                              float(g('min_intensity', 0.01)), int(max_peaks), scaling,
                              int(g('min_peaks', 10)), float(g('min_mz_range', 250.0)),
                              int(resolution is not None), int(resolution or 0))
    r = raw.to(device).contiguous()
    n = r.n
    dev = r.mz.device
    o_mz = torch.zeros((n, max_peaks), dtype=torch.float32, device=dev)
    o_in = torch.zeros((n, max_peaks), dtype=torch.float32, device=dev)
    o_src = torch.zeros((n, max_peaks), dtype=torch.int32, device=dev)
    o_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    o_val = torch.zeros(n, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().asl_process_batch(C.byref(_lib.peaks_struct(r)), C.byref(P),
                                            _lib.ptr(o_mz), _lib.ptr(o_in), _lib.ptr(o_src),
                                            _lib.ptr(o_cnt), _lib.ptr(o_val)))
    slot = torch.arange(max_peaks, device=dev).expand(n, -1) < o_cnt.unsqueeze(1)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(o_cnt.to(torch.int64), 0)
    src_abs = (o_src.to(torch.int64) + r.offsets[:-1].to(torch.int64).unsqueeze(1))[slot]
    out = PackedSpectra(offsets.to(torch.int32), o_mz[slot], o_in[slot], r.charge[src_abs],
                        r.precursor_mz.clone(), r.precursor_charge.clone(),
                        identifiers=raw.identifiers)
    return out, o_val.bool()


@dataclass
class SpectrumSpectrumMatch:
    """The fields of the reference's SSM the writer consumes (spectrum.py:217-271,
    writer.py:129-148), flattened."""
    sequence: str
    query_identifier: str
    query_index: int
    library_identifier: Any
    retention_time: Any
    charge: int
    exp_mass_to_charge: float
    calc_mass_to_charge: float
    is_decoy: bool
    search_engine_score: float = float('nan')
    q: float = float('nan')
    peak_matches: Optional[np.ndarray] = None


def ssms_from_batch(result, query_meta, library_meta, scores=None, q_values=None
                    ) -> List[SpectrumSpectrumMatch]:
    """SSMs of one ``BatchResult``. ``query_meta[i]`` / ``library_meta[row]`` are mappings with
    the reference's attribute names (identifier, index, retention_time, precursor_charge,
    precursor_mz / identifier, peptide, precursor_mz, is_decoy). Queries without a
    candidate are skipped (spectral_library.py:359)."""
    out = []
    for i in range(len(result.best_row)):
        r = int(result.best_row[i])
        if r < 0:
            continue
        qm, lm = query_meta[i], library_meta[r]
        out.append(SpectrumSpectrumMatch(
            lm['peptide'], qm['identifier'], qm['index'], lm['identifier'],
            qm.get('retention_time'), qm['precursor_charge'], qm['precursor_mz'],
            lm['precursor_mz'], lm.get('is_decoy', False),
            float('nan') if scores is None else scores[i],
            float('nan') if q_values is None else q_values[i], result.peak_matches(i)))
    return out

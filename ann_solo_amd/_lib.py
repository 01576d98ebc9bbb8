"""ctypes loader for libannsolo_mi.so (the C ABI of include/annsolo_mi.h).

The product path has no CPU fallback: if the shared library is missing or no HIP
device is present, every compute call raises ``AnnSoloMiError``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASL_LIB_PATH: load another build of the same library (same-box A/B measurements)
LIB_PATH = os.environ.get('ASL_LIB_PATH') or os.path.join(_HERE, 'libannsolo_mi.so')
_lib = None

c_f32p = C.POINTER(C.c_float)
c_f64p = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)


TK_MAX_K = 2048      # largest nprobe / single-pass k of the LDS top-k (csrc/ivf_kernels.hpp)
TK_MAX_K_PASSES = 16384   # largest k of a search (bounded passes beyond TK_MAX_K)


class AnnSoloMiError(RuntimeError):
    pass


class AslPeaks(C.Structure):
    _fields_ = [('n', C.c_int32), ('offsets', C.c_void_p), ('mz', C.c_void_p),
                ('intensity', C.c_void_p), ('charge', C.c_void_p),
                ('precursor_mz', C.c_void_p), ('precursor_charge', C.c_void_p),
                ('n_peaks', C.c_int64)]


class AslSearchParams(C.Structure):
    _fields_ = [('min_bound', C.c_double), ('bin_size', C.c_double),
                ('hash_seed', C.c_uint32), ('k', C.c_int32), ('nprobe', C.c_int32),
                ('charge', C.c_int32), ('precursor_tol', C.c_double),
                ('precursor_mode', C.c_int32), ('fragment_mz_tolerance', C.c_double),
                ('allow_shift', C.c_int32), ('use_ann', C.c_int32)]


class AslProcessParams(C.Structure):
    _fields_ = [('min_mz', C.c_double), ('max_mz', C.c_double), ('remove_precursor', C.c_int32),
                ('remove_precursor_tolerance', C.c_double), ('min_intensity', C.c_double),
                ('max_peaks', C.c_int32), ('scaling', C.c_int32), ('min_peaks', C.c_int32),
                ('min_mz_range', C.c_double), ('round_mz', C.c_int32), ('resolution', C.c_int32)]


class AslIndexInfo(C.Structure):
    _fields_ = [('d', C.c_int32), ('nlist', C.c_int32), ('kind', C.c_int32),
                ('pq_m', C.c_int32), ('pq_ksub', C.c_int32), ('pq_dsub', C.c_int32),
                ('ntotal', C.c_int64), ('nlocal', C.c_int64), ('trained', C.c_int32),
                ('shard_rank', C.c_int32), ('shard_world', C.c_int32)]


EXPORTS = [
    'asl_last_error', 'asl_version', 'asl_get_num_gpus', 'asl_set_device', 'asl_set_stream',
    'asl_synchronize', 'asl_set_pipeline', 'asl_set_scan_postfilter', 'asl_get_dim', 'asl_hash_idx', 'asl_encode_batch', 'asl_index_create',
    'asl_index_free', 'asl_index_train', 'asl_index_add', 'asl_index_add_preassigned', 'asl_index_search',
    'asl_index_reset', 'asl_index_ntotal', 'asl_index_is_trained', 'asl_index_save',
    'asl_index_load', 'asl_index_set_niter', 'asl_index_info', 'asl_index_get_centroids',
    'asl_index_get_codebooks', 'asl_index_set_trained', 'asl_index_get_lists',
    'asl_index_shard', 'asl_index_shard_map', 'asl_topk_merge', 'asl_index_coarse',
    'asl_index_pq_lut', 'asl_rescore_batch', 'asl_library_create', 'asl_library_free',
    'asl_library_size', 'asl_search_batch', 'asl_window_candidates', 'asl_profile_enable',
    'asl_profile_reset', 'asl_profile_get', 'asl_profile_scanned_vectors',
    'asl_rescore_knn', 'asl_lpt_owner', 'asl_index_supports_keys', 'asl_index_search_sharded', 'asl_index_postings_work', 'asl_index_set_refine', 'asl_index_get_refine', 'asl_index_refine', 'asl_index_set_scan_variant', 'asl_index_search_preassigned', 'asl_process_batch',
    'asl_ssm_features_batch', 'asl_ssm_cosine_batch', 'asl_index_set_unordered', 'asl_topk_merge_keys',
    'asl_index_set_flat_storage', 'asl_index_get_flat_storage', 'asl_index_flat_layout',
    'asl_keys_split', 'asl_keys_merge_heads', 'asl_keys_extras', 'asl_keys_merge_final',
    'asl_keys_rescan_list', 'asl_shard_k', 'asl_index_search_gated', 'asl_index_search_entries',
    'asl_encode_entries_batch', 'asl_index_search_sharded_ex',
]


def build(force: bool = False) -> str:
    """Compile libannsolo_mi.so for gfx950 with hipcc (cross-compiles without a GPU).
    ``force`` (or ASL_FORCE_REBUILD=1) recompiles every translation unit; otherwise only what is
    older than its sources. Every run appends one line to ``csrc/build/build.log`` naming the
    objects that were recompiled."""
    import time
    src = os.path.join(_HERE, 'csrc')
    force = force or os.environ.get('ASL_FORCE_REBUILD', '') not in ('', '0')
    stale = force or not os.path.exists(LIB_PATH)
    if not stale:
        t = os.path.getmtime(LIB_PATH)
        deps = [os.path.join(src, f) for f in os.listdir(src)
                if f.endswith(('.hip', '.hpp'))]
        deps.append(os.path.join(_HERE, '..', 'include', 'annsolo_mi.h'))
        stale = any(os.path.getmtime(p) > t for p in deps)
    rebuilt = []
    if stale:
        if not os.path.exists('/opt/rocm/bin/hipcc'):
            raise AnnSoloMiError('hipcc not found and libannsolo_mi.so is stale/missing')
        obj_dir = os.path.join(src, 'build')
        before = {f: os.path.getmtime(os.path.join(obj_dir, f))
                  for f in (os.listdir(obj_dir) if os.path.isdir(obj_dir) else []) if f.endswith('.o')}
        cmd = ['make', '-C', src, '-j8'] + (['-B'] if force else [])
        subprocess.check_call(cmd)
        for f in sorted(os.listdir(obj_dir)):
            if f.endswith('.o') and os.path.getmtime(os.path.join(obj_dir, f)) > before.get(f, 0.0):
                rebuilt.append(f)
    try:
        os.makedirs(os.path.join(src, 'build'), exist_ok=True)
        with open(os.path.join(src, 'build', 'build.log'), 'a') as f:
            f.write('%s force=%d recompiled=%d [%s] -> %s\n' % (
                time.strftime('%Y-%m-%dT%H:%M:%S'), int(force), len(rebuilt), ' '.join(rebuilt),
                os.path.basename(LIB_PATH)))
    except OSError:
        pass
    print('[build] libannsolo_mi.so: %d translation units recompiled for gfx950%s'
          % (len(rebuilt), (' (' + ' '.join(rebuilt) + ')') if rebuilt else ' (up to date)'))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AnnSoloMiError(
                f'{LIB_PATH} is missing: build it with __graft_entry__.build() '
                '(there is no CPU fallback)')
        L = C.CDLL(LIB_PATH)
        L.asl_last_error.restype = C.c_char_p
        L.asl_version.restype = C.c_char_p
        L.asl_hash_idx.restype = C.c_int32
        L.asl_hash_idx.argtypes = [C.c_int64, C.c_int32, C.c_uint32]
        L.asl_get_dim.argtypes = [C.c_double, C.c_double, C.c_double, c_i64p, c_f64p, c_f64p]
        L.asl_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_double, C.c_double, C.c_int32, C.c_uint32, C.c_int,
                                       C.c_void_p]
        L.asl_index_create.restype = C.c_void_p
        L.asl_index_create.argtypes = [C.c_int32] * 5
        L.asl_index_free.argtypes = [C.c_void_p]
        L.asl_index_free.restype = None
        L.asl_index_train.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64]
        L.asl_index_add.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.asl_index_add_preassigned.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.asl_index_search.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p]
        L.asl_index_search_preassigned.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                                   C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p]
        L.asl_process_batch.argtypes = [C.POINTER(AslPeaks), C.POINTER(AslProcessParams),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.asl_ssm_features_batch.argtypes = [C.POINTER(AslPeaks), C.POINTER(AslPeaks), C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int32, C.c_double,
                                             C.c_double, C.c_double, C.c_int32, C.c_void_p]
        L.asl_ssm_cosine_batch.argtypes = [C.POINTER(AslPeaks), C.POINTER(AslPeaks), C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.asl_index_search_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                               C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.asl_index_search_sharded_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                                  C.c_int32, C.c_int32, C.c_int64]
        L.asl_index_set_refine.argtypes = [C.c_void_p, C.c_int32]
        L.asl_index_refine.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_int32, C.c_void_p, C.c_void_p]
        L.asl_index_reset.argtypes = [C.c_void_p]
        L.asl_index_ntotal.argtypes = [C.c_void_p]
        L.asl_index_ntotal.restype = C.c_int64
        L.asl_index_is_trained.argtypes = [C.c_void_p]
        L.asl_index_save.argtypes = [C.c_void_p, C.c_char_p]
        L.asl_index_load.argtypes = [C.c_char_p]
        L.asl_index_load.restype = C.c_void_p
        L.asl_index_set_niter.argtypes = [C.c_void_p, C.c_int32]
        L.asl_index_set_scan_variant.argtypes = [C.c_void_p, C.c_int32]
        L.asl_index_set_flat_storage.argtypes = [C.c_void_p, C.c_int32]
        L.asl_index_flat_layout.argtypes = [C.c_void_p]
        L.asl_index_get_flat_storage.argtypes = [C.c_void_p]
        L.asl_keys_split.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.asl_keys_merge_heads.argtypes = [C.c_int32] * 4 + [C.c_void_p] * 4
        L.asl_keys_rescan_list.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        L.asl_keys_extras.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int32]
        L.asl_keys_merge_final.argtypes = [C.c_int32] * 4 + [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                                             C.c_void_p, C.c_void_p, C.c_void_p]
        L.asl_shard_k.argtypes = [C.c_int32, C.c_int32]
        L.asl_shard_k.restype = C.c_int32
        L.asl_index_supports_keys.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.asl_index_search_gated.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.asl_index_search_entries.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.asl_encode_entries_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                               C.c_double, C.c_double, C.c_int32, C.c_uint32, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p]
        L.asl_index_postings_work.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, c_i64p, c_i64p]
        L.asl_index_get_refine.argtypes = [C.c_void_p]
        L.asl_index_get_refine.restype = C.c_int32
        L.asl_index_set_unordered.argtypes = [C.c_void_p, C.c_int32]
        L.asl_topk_merge_keys.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int32]
        L.asl_index_info.argtypes = [C.c_void_p, C.POINTER(AslIndexInfo)]
        L.asl_index_get_centroids.argtypes = [C.c_void_p, C.c_void_p]
        L.asl_index_get_codebooks.argtypes = [C.c_void_p, C.c_void_p]
        L.asl_index_set_trained.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.asl_index_get_lists.argtypes = [C.c_void_p] * 5
        L.asl_index_shard.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.asl_index_shard_map.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.asl_topk_merge.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
        L.asl_index_coarse.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p]
        L.asl_index_pq_lut.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.asl_rescore_batch.argtypes = [C.POINTER(AslPeaks), C.POINTER(AslPeaks), C.c_void_p,
                                        C.c_void_p, C.c_double, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.asl_library_create.restype = C.c_void_p
        L.asl_library_create.argtypes = [C.POINTER(AslPeaks), C.c_void_p, C.c_void_p]
        L.asl_library_free.argtypes = [C.c_void_p]
        L.asl_library_free.restype = None
        L.asl_library_size.argtypes = [C.c_void_p]
        L.asl_library_size.restype = C.c_int64
        L.asl_search_batch.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(AslPeaks),
                                       C.POINTER(AslSearchParams), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p]
        L.asl_rescore_knn.argtypes = [C.c_void_p, C.POINTER(AslPeaks),
                                      C.POINTER(AslSearchParams), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.asl_lpt_owner.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        L.asl_window_candidates.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                            C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
        L.asl_set_stream.argtypes = [C.c_void_p]
        L.asl_profile_get.argtypes = [C.c_char_p, c_f64p, c_i64p]
        L.asl_profile_scanned_vectors.restype = C.c_int64
        _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        msg = lib().asl_last_error()
        raise AnnSoloMiError(f'libannsolo_mi error {rc}: {msg.decode() if msg else ""}')


def ptr(a):
    """Raw address of a numpy array or torch tensor (host or device), or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if not a.flags['C_CONTIGUOUS']:
            raise ValueError('array must be C-contiguous')
        return a.ctypes.data
    if hasattr(a, 'data_ptr'):
        if not a.is_contiguous():
            raise ValueError('tensor must be contiguous')
        if a.is_cuda:
            _require_default_stream(a.device)
        return a.data_ptr()
    raise TypeError(type(a))


_DEFAULT_RAW = {}


def _default_raw_stream(idx: int) -> int:
    import torch
    if idx not in _DEFAULT_RAW:
        _DEFAULT_RAW[idx] = int(torch.cuda.default_stream(idx).cuda_stream)
    return _DEFAULT_RAW[idx]


def _require_default_stream(device):
    """The library issues its kernels on the null stream (and, in pipeline mode, on streams of
    its own that it orders against the null stream). That is ordered with PyTorch only while
    PyTorch's CURRENT stream is its default stream: under ``torch.cuda.stream(s)`` with a
    non-blocking ``s`` the tensors handed over here could still be written, or be read back
    before the library has finished. Every device tensor crosses ``ptr()``, so the check lives
    here and such a call fails instead of racing."""
    import torch
    # (every device tensor of every call passes here: the raw-handle query is a C call of well
    # under a microsecond; the stream objects of the public API cost ~10 us a pair)
    try:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        ok = torch._C._cuda_getCurrentRawStream(idx) == _default_raw_stream(idx)
    except AttributeError:      # a PyTorch without the private hook
        ok = torch.cuda.current_stream(device) == torch.cuda.default_stream(device)
    if not ok:
        raise AnnSoloMiError(
            'libannsolo_mi issues its work on the default HIP stream: call it with '
            "PyTorch's default stream current (not inside torch.cuda.stream(...)), or "
            'synchronise the side stream and switch back first')


def peaks_struct(p) -> AslPeaks:
    """AslPeaks view of a PackedSpectra (torch) or a tuple of numpy arrays."""
    if hasattr(p, 'offsets') and hasattr(p, 'precursor_charge'):
        arrs = (p.offsets, p.mz, p.intensity, p.charge, p.precursor_mz, p.precursor_charge)
        n = p.n
    else:
        arrs = p
        n = len(p[0]) - 1
    # a pack's peak arrays are exactly sized (offsets[n] == len(mz)): telling the library spares
    # it the read-back of offsets[n] from the device (a stream synchronisation per call)
    mz = arrs[1]
    n_peaks = int(mz.numel()) if hasattr(mz, 'numel') else int(len(mz))
    return AslPeaks(n, *[ptr(a) for a in arrs], n_peaks)

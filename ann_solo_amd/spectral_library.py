"""Search engine -- host-side mirror of the hot-path half of the reference's
``ann_solo/spectral_library.py`` (``SpectralLibrary`` :27-500).

What is mirrored (same names, same meaning): the constructor (:46-116: ``SpectralLibrary(
filename)`` over a ``SpectralLibraryReader``, kept as ``_library_reader``),
``_get_hyperparameter_hash`` (:118-131), ``_create_ann_indexes`` (:133-183), ``search`` /
``_search_cascade`` (:193-326), ``_search_batch`` (:328-370), ``_get_library_candidates``
(:372-455), ``_get_ann_index`` (:457-500) and ``shutdown`` (:185-191). What is different by
design: the library is a packed, HBM-resident peak store per precursor charge (built once
through the reader's surface, ``library_store.py``) instead of per-spectrum HDF5 reads, a
whole batch runs through ``asl_search_batch`` in one device pipeline (encode -> IVF top-k ->
precursor post-filter -> shifted-dot best match), and with ``enable_sharding`` every batch of
both cascade levels runs over the ranks of one node (``distributed.py``).

File parsing, FDR/mokapot scoring stay with the reference (out of scope): the
library/query readers and the scorer are injected (defaults: the reference's own
``ann_solo.reader`` when it is importable).
"""
import ctypes as C
import hashlib
import json
import logging
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import faiss_compat as faiss
from .config import Config
from .packed import PackedSpectra
from .spectrum import get_dim, spectra_to_vectors, HASH_SEED


@dataclass
class ChargePartition:
    """spec_info['charge'][z] (reader.py:184-191) + the device-resident peak store."""
    charge: int
    ids: np.ndarray                 # library identifiers of the rows
    precursor_mz: np.ndarray        # float32, as the reference stores it
    spectra: PackedSpectra          # device
    handle: C.c_void_p = None       # asl_library_t*
    index: Optional[faiss.Index] = None


@dataclass
class BatchResult:
    """Per-query outputs of one batch (all numpy, length nq)."""
    best_row: np.ndarray            # row inside the charge partition, -1: no candidate
    best_score: np.ndarray
    n_candidates: np.ndarray
    pm_count: np.ndarray
    pm_pairs: np.ndarray            # [nq, stride, 2]
    knn: Optional[np.ndarray] = None

    def peak_matches(self, i) -> np.ndarray:
        return self.pm_pairs[i, :self.pm_count[i]].astype(np.int64)


INDEX_EXT = '.idxmi'      # own container; the reference's FAISS files keep '.idxann' untouched


def _reference_reader_factory(filename: str, config_hash: str):
    """Default library reader: the reference's own ``SpectralLibraryReader`` (reader.py:40-116)
    when the ``ann_solo`` package is installed next to this one."""
    try:
        from ann_solo import reader as ref_reader          # noqa: the reference package
    except Exception as e:                                   # faiss/h5py/... missing
        raise ImportError(
            'SpectralLibrary(filename) needs a library reader: install the reference package '
            '(ann_solo.reader.SpectralLibraryReader) or pass reader_factory=') from e
    return ref_reader.SpectralLibraryReader(filename, config_hash)


def _reference_config():
    """Default options: the reference's parsed ``config`` singleton (``ann_solo.py:76``) when the
    ``ann_solo`` package is installed and ``config.parse()`` has run, else the defaults."""
    try:
        from ann_solo.config import config as ref_config      # noqa: the reference package
    except Exception:
        return None
    return ref_config if getattr(ref_config, '_namespace', None) is not None else None


def _reference_query_reader(filename: str):
    try:
        from ann_solo import reader as ref_reader
    except Exception as e:
        raise ImportError('search(query_filename) needs a query reader: install the reference '
                          'package (ann_solo.reader.read_query_file) or pass query_reader=') from e
    return ref_reader.read_query_file(filename)


class SpectralLibrary:
    _hyperparameters = ['min_mz', 'max_mz', 'bin_size', 'hash_len', 'num_list']

    def __init__(self, library, identifiers=None, config: Config = None,
                 valid: Optional[np.ndarray] = None, index_dir: Optional[str] = None,
                 basename: Optional[str] = None, device='cuda', reader_factory=None,
                 query_reader=None, score_ssms=None, annotation_alignment: str = 'peaks',
                 import_faiss_cache: bool = True):
        """``library`` is one of

        * a file name -- the reference's call, ``SpectralLibrary(filename)``
          (spectral_library.py:46-116): ``reader_factory(filename, hyper_hash)`` (default: the
          reference's ``SpectralLibraryReader``) opens it, the packed store and the ANN indexes
          are cached next to it as ``<library>_<hash7>.spstore`` / ``<library>_<hash7>_<charge>
          .idxmi``;
        * a reader object with the reference reader's surface (``spec_info``,
          ``read_all_spectra()``, optional ``is_recreated`` / ``get_version()`` / ``close()``);
        * a ``PackedSpectra`` of already processed library spectra (``identifiers``: library
          ids, default row numbers; ``valid``: is_valid flags).

        ANN indexes are built for every charge with >= num_list spectra (:100-104) and cached
        under ``index_dir`` when one is known (:98-108). ``query_reader(filename)`` (default: the
        reference's ``read_query_file``) and ``score_ssms(ssms, mode)`` (stands for
        ``utils.score_ssms``, :319-326) serve ``search(query_filename)``.
        ``annotation_alignment='snapshot'`` reproduces the snapshot's use of RAW-peak
        annotations on processed library peaks (reader.py:243-245; see library_store.py).
        ``import_faiss_cache``: with the reference's own index type (``index='ivfflat'``) an
        existing ``<library>_<hash7>_<charge>.idxann`` written by the reference's FAISS is loaded
        (its centroids and list assignments kept) instead of training a new index."""
        self.config = Config.from_reference(config if config is not None
                                            else _reference_config())
        self.device = torch.device(device)
        cfg = self.config
        if cfg.no_gpu:
            # the reference's --no_gpu keeps FAISS on the CPU (spectral_library.py:73-75); this
            # engine IS the GPU path and has no CPU fallback: say so instead of ignoring the flag
            raise _lib.AnnSoloMiError(
                'no_gpu=True: ann_solo_amd is the MI355X path and has no CPU fallback; run the '
                "reference's own SpectralLibrary for a CPU search")
        k_max = _lib.TK_MAX_K
        if cfg.num_probe > k_max:
            # the reference clamps both to 1024 on the GPU (FAISS-GPU's limit, :76-87); this
            # implementation's LDS top-k holds 2048 -- same warning, the larger limit
            logging.warning('Using num_probe=%d (maximum supported value on the GPU), %d was '
                            'supplied', k_max, cfg.num_probe)
        c_max = _lib.TK_MAX_K_PASSES       # beyond k_max the index searches in bounded passes (exact, slower)
        if cfg.num_candidates > c_max:
            logging.warning('Using num_candidates=%d (maximum supported value on the GPU), %d '
                            'was supplied', c_max, cfg.num_candidates)
        self._num_probe = min(cfg.num_probe, k_max)
        self._num_candidates = min(cfg.num_candidates, c_max)
        self._use_gpu = True
        self._ann_filenames: Dict[int, str] = {}
        self._faiss_filenames: Dict[int, str] = {}      # reference caches that can be imported
        self._current_index: Tuple[Optional[int], Optional[faiss.Index]] = (None, None)
        self._query_reader = query_reader or _reference_query_reader
        self._score_ssms = score_ssms
        self._dist = None
        self._library_reader = None
        self.library_meta: Optional[Dict[int, list]] = None
        self.partitions: Dict[int, ChargePartition] = {}
        verify_file_existence = True
        if isinstance(library, PackedSpectra):
            self._index_dir = index_dir
            self._basename = basename or 'library'
            self._init_from_packed(library, identifiers, valid)
        else:
            from . import library_store as ls
            if isinstance(library, (str, os.PathLike)):
                filename = os.fspath(library)
                self._library_reader = (reader_factory or _reference_reader_factory)(
                    filename, self._get_hyperparameter_hash())
            else:
                self._library_reader = library
                filename = getattr(library, '_filename', None) or cfg.spectral_library_filename
            if filename:
                stem = os.path.splitext(filename)[0]
                self._index_dir = index_dir if index_dir is not None else (os.path.dirname(stem) or '.')
                self._basename = basename or os.path.basename(stem)
            else:
                self._index_dir, self._basename = index_dir, basename or 'library'
            if getattr(self._library_reader, 'is_recreated', False):
                logging.warning('ANN indexes were created using non-compatible settings')
                verify_file_existence = False
            key = ls.store_hash(cfg, self._get_hyperparameter_hash(), annotation_alignment)
            path = None if self._index_dir is None else os.path.join(
                self._index_dir, f'{self._basename}_{key[:7]}{ls.STORE_EXT}')
            store = ls.load_or_build_library_store(self._library_reader, cfg, self.device, path,
                                                   key, annotation_alignment)
            self.library_meta = store.meta
            for z, (a, b) in store.ranges.items():
                rows = torch.arange(a, b)
                info = self._library_reader.spec_info['charge'][z]
                self._add_partition(int(z), store.spectra.select(rows), np.asarray(info['id']),
                                    store.valid[a:b],
                                    np.asarray(info['precursor_mz'], np.float32))
        if cfg.mode == 'ann':
            create = []
            for z in sorted(self.partitions):
                if len(self.partitions[z].ids) < cfg.num_list:
                    continue          # infrequent charge: brute force (spectral_library.py:102-104)
                base = f'{self._basename}_{self._get_index_hash()[:7]}'
                self._ann_filenames[z] = os.path.join(self._index_dir or '', f'{base}_{z}{INDEX_EXT}')
                ref = os.path.join(self._index_dir or '', f'{self._basename}_'
                                   f'{self._get_hyperparameter_hash()[:7]}_{z}.idxann')
                if (import_faiss_cache and cfg.index == 'ivfflat' and self._index_dir is not None and
                        verify_file_existence and os.path.isfile(ref)):
                    self._faiss_filenames[z] = ref
                if (self._index_dir is None or not verify_file_existence or
                        not (os.path.isfile(self._ann_filenames[z]) or z in self._faiss_filenames)):
                    if self._index_dir is not None:
                        logging.warning('Missing ANN index for charge %d', z)
                    create.append(z)
            if create:
                self._create_ann_indexes(create)
        if cfg.num_gpus and cfg.num_gpus > 1:      # additive flag --num_gpus (config.py)
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_initialized() else 1
            if world != cfg.num_gpus:
                raise RuntimeError(
                    f'num_gpus={cfg.num_gpus} needs a torch.distributed job of that many ranks '
                    f'(one process per GPU, e.g. torchrun --nproc-per-node {cfg.num_gpus}); '
                    f'this process sees {world}')
            self.enable_sharding()

    def _init_from_packed(self, library: PackedSpectra, identifiers, valid) -> None:
        n = library.n
        ids = np.arange(n) if identifiers is None else np.asarray(identifiers)
        pz = library.precursor_charge.cpu().numpy()
        valid = np.ones(n, bool) if valid is None else np.asarray(valid, bool)
        for z in np.unique(pz):
            rows = np.nonzero(pz == z)[0]
            self._add_partition(int(z), library.select(torch.as_tensor(rows)), ids[rows],
                                valid[rows], None)

    def _add_partition(self, z: int, spectra: PackedSpectra, ids, valid, pmz32) -> None:
        part = spectra.to(self.device).contiguous()
        if pmz32 is None:       # spec_info stores the precursor m/z column as float32 (reader.py:186-189)
            pmz32 = part.precursor_mz.cpu().numpy().astype(np.float32)
        pmz32 = np.ascontiguousarray(pmz32, np.float32)
        v = np.ascontiguousarray(np.asarray(valid).astype(np.uint8))
        h = _lib.lib().asl_library_create(C.byref(_lib.peaks_struct(part)), _lib.ptr(pmz32),
                                          _lib.ptr(v))
        if not h:
            _lib.check(-1)
        self.partitions[z] = ChargePartition(z, np.asarray(ids), pmz32, part, C.c_void_p(h))

    # ------------------------------------------------------------------ reference mirrors
    def _get_hyperparameter_hash(self) -> str:
        b = json.dumps({hp: self.config[hp] for hp in self._hyperparameters}).encode('utf-8')
        return hashlib.sha1(b).hexdigest()

    def _get_index_hash(self) -> str:
        """Hash in the cached index file names: the reference's five hyper-parameters for its
        own configuration (IVF-Flat, FAISS' default 25 iterations and seed 1234 -- same 7 hex
        digits as the reference's ``.idxann`` name); any additive option of this implementation
        (index kind, PQ shape, trainer settings) is hashed in as well, so switching it can never
        pick up a stale file."""
        cfg = self.config
        if (cfg.index == 'ivfflat' and cfg.kmeans_niter == 25 and cfg.seed == 1234 and
                cfg.flat_storage == 'fp32'):
            return self._get_hyperparameter_hash()
        d = {hp: cfg[hp] for hp in self._hyperparameters}
        d.update(index=cfg.index, kmeans_niter=cfg.kmeans_niter, seed=cfg.seed)
        if cfg.index == 'ivfflat' and cfg.flat_storage != 'fp32':
            d.update(flat_storage=cfg.flat_storage)
        if cfg.index == 'ivfpq':
            d.update(pq_m=cfg.pq_m, pq_bits=cfg.pq_bits)
            if cfg.refine_k:
                d.update(refine_k=cfg.refine_k)
        return hashlib.sha1(json.dumps(d).encode('utf-8')).hexdigest()

    def _encode(self, spectra: PackedSpectra) -> torch.Tensor:
        cfg = self.config
        out = torch.empty((spectra.n, cfg.hash_len), dtype=torch.float32, device=self.device)
        spectra_to_vectors(spectra.mz, spectra.intensity, spectra.offsets, cfg.min_mz, cfg.max_mz,
                           cfg.bin_size, cfg.hash_len, True, out)
        return out

    def _create_ann_indexes(self, charges: List[int]) -> None:
        cfg = self.config
        for z in charges:
            part = self.partitions[z]
            vectors = self._encode(part.spectra)
            quantizer = faiss.IndexFlatIP(cfg.hash_len)
            if cfg.index == 'ivfpq':
                ann_index = faiss.IndexIVFPQ(quantizer, cfg.hash_len, cfg.num_list, cfg.pq_m,
                                             cfg.pq_bits, faiss.METRIC_INNER_PRODUCT)
            else:
                ann_index = faiss.IndexIVFFlat(quantizer, cfg.hash_len, cfg.num_list,
                                               faiss.METRIC_INNER_PRODUCT, storage=cfg.flat_storage)
            ann_index.seed = cfg.seed
            ann_index.set_niter(cfg.kmeans_niter)
            if cfg.index == 'ivfpq' and cfg.refine_k:
                ann_index.set_refine(cfg.refine_k)
            ann_index.train(vectors)
            ann_index.add(vectors)
            if self._index_dir is not None:
                faiss.write_index(ann_index, self._ann_filenames[z])
            part.index = ann_index
            del vectors

    def _index_matches(self, idx: faiss.Index, part: ChargePartition) -> bool:
        cfg, i = self.config, idx.info()
        kind = 2 if cfg.index == 'ivfpq' else 1
        return (i.kind == kind and i.d == cfg.hash_len and i.nlist == cfg.num_list and
                i.ntotal == len(part.ids) and bool(i.trained) and i.shard_world == 1 and
                (kind != 2 or (i.pq_m == cfg.pq_m and i.pq_ksub == (1 << cfg.pq_bits))) and
                (kind != 1 or idx.storage == cfg.flat_storage))

    def _get_ann_index(self, charge: int) -> faiss.Index:
        part = self.partitions[charge]
        if part.index is None:
            idx = None
            imported = False
            if not os.path.isfile(self._ann_filenames[charge]) and charge in self._faiss_filenames:
                try:        # the reference's FAISS cache: its centroids, its list assignments
                    idx = faiss.read_index_faiss(self._faiss_filenames[charge],
                                                 storage=self.config.flat_storage)
                    imported = True
                    logging.info('Imported the FAISS index %s', self._faiss_filenames[charge])
                except (ValueError, OSError, _lib.AnnSoloMiError) as e:
                    logging.warning('FAISS index %s not usable (%s): building a new index',
                                    self._faiss_filenames[charge], e)
            else:
                try:
                    idx = faiss.read_index(self._ann_filenames[charge])
                except _lib.AnnSoloMiError as e:
                    logging.warning('ANN index %s unreadable (%s): rebuilding',
                                    self._ann_filenames[charge], e)
            if idx is not None and not self._index_matches(idx, part):
                # e.g. another library under the same base name: out-of-range ids would be
                # dropped silently by the rescoring bounds check -- never search a stale index
                logging.warning('ANN index %s does not match the library/configuration: '
                                'rebuilding', self._ann_filenames[charge])
                idx = None
            if idx is None:
                self._create_ann_indexes([charge])
            else:
                part.index = idx
                if imported and self._index_dir is not None:     # next time: our own container
                    faiss.write_index(idx, self._ann_filenames[charge])
            d = self._dist
            if d is not None and d.world > 1:
                part.index.shard(d.rank, d.world)
        part.index.nprobe = self._num_probe
        self._current_index = charge, part.index
        return part.index

    # ------------------------------------------------------------------ multi-GPU
    def enable_sharding(self, group=None) -> None:
        """List-shard every ANN index over the ranks of ``group`` (default: all ranks of the
        initialised ``torch.distributed`` job) -- SURVEY.md 8(e). From here on every batch of
        both cascade levels runs on all ranks (``distributed.sharded_cascade_batch``): the open
        search scans this rank's inverted lists for the whole batch and exchanges per-shard
        top-k, the standard search is data-parallel over the queries; every rank ends up with
        the results of the whole batch. Every rank must call ``search`` with the same queries."""
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self._dist is not None and self._dist.world > 1:
            raise RuntimeError('enable_sharding: the indexes are sharded already')
        self._dist = SimpleNamespace(group=group, world=world, rank=rank, backends={})
        if world > 1:
            for part in self.partitions.values():
                if part.index is not None:
                    part.index.shard(rank, world)

    def set_pipeline(self, on=True) -> None:
        """Two-stream software pipeline of the device hot path (``asl_set_pipeline``): with
        ``device_out=True`` an open-search ``_search_batch`` returns without waiting, and the
        encoder + coarse quantiser of the next batch run under the list scan of this one.
        Results are valid after ``synchronize()``; values are bit-identical either way."""
        self.synchronize()
        _lib.check(_lib.lib().asl_set_pipeline(int(on)))       # True/1: two streams, 3: three
        self._pipeline_on = bool(on)

    def synchronize(self) -> None:
        """Wait for the batches in flight; their inputs and outputs may be released afterwards."""
        try:
            _lib.check(_lib.lib().asl_synchronize())
        finally:
            getattr(self, '_inflight', []).clear()

    def _hold(self, *tensors) -> None:
        """Pipeline mode returns before the device has read the inputs or written the outputs,
        on streams PyTorch's allocator knows nothing about: a tensor released by the caller could be
        handed out again (and overwritten) while a kernel still uses it. Every array of a pipelined
        call is therefore kept alive here until the next synchronisation."""
        if not hasattr(self, '_inflight'):
            self._inflight = []
        self._inflight.append(tensors)
        if len(self._inflight) >= 32:      # bound what is pinned: waiting here is cheap
            self.synchronize()

    def _shard_backend(self, charge: int, mode: str):
        from .distributed import HipShardBackend
        key = (charge, mode)
        if key not in self._dist.backends:
            self._dist.backends[key] = HipShardBackend(self, charge, mode)
        return self._dist.backends[key]

    def shutdown(self) -> None:
        if getattr(self, '_inflight', None):
            self.synchronize()
        reader = getattr(self, '_library_reader', None)
        if reader is not None and hasattr(reader, 'close'):
            reader.close()                       # spectral_library.py:189
        for part in self.partitions.values():
            if part.handle:
                _lib.lib().asl_library_free(part.handle)
                part.handle = None
            part.index = None

    def _tolerance(self, mode: str):
        cfg = self.config
        if mode == 'std':
            return cfg.precursor_tolerance_mass, cfg.precursor_tolerance_mode
        if mode == 'open':
            return cfg.precursor_tolerance_mass_open, cfg.precursor_tolerance_mode_open
        raise ValueError('Unknown search mode')

    def _uses_ann(self, charge: int, mode: str) -> bool:
        return self.config.mode == 'ann' and mode == 'open' and charge in self._ann_filenames

    def _search_batch(self, queries: PackedSpectra, charge: int, mode: str,
                      want_knn: bool = False, device_out: bool = False) -> Optional[BatchResult]:
        """One batch of same-charge, processed query spectra (spectral_library.py:328-370):
        through the device hot path of this GPU, or -- after ``enable_sharding`` on more than
        one rank -- over all ranks. Returns None when the library has no spectra of that charge
        (:411-412)."""
        d = getattr(self, '_dist', None)
        if d is None or d.world == 1:
            return self._search_batch_local(queries, charge, mode, want_knn, device_out)
        tol_val, tol_mode = self._tolerance(mode)
        if tol_mode not in ('Da', 'ppm'):
            raise ValueError('Unknown precursor tolerance mode')
        if charge not in self.partitions or queries.n == 0:
            return None if charge not in self.partitions else self._search_batch_local(
                queries, charge, mode)
        from .distributed import sharded_cascade_batch
        return sharded_cascade_batch(self._shard_backend(charge, mode), queries, mode,
                                     self._uses_ann(charge, mode), d.group)

    def _search_batch_local(self, queries: PackedSpectra, charge: int, mode: str,
                            want_knn: bool = False, device_out: bool = False,
                            pm_stride: Optional[int] = None) -> Optional[BatchResult]:
        """One batch on this GPU alone (the index must not be sharded for open searches)."""
        tol_val, tol_mode = self._tolerance(mode)
        if tol_mode not in ('Da', 'ppm'):
            raise ValueError('Unknown precursor tolerance mode')
        if charge not in self.partitions:
            return None
        cfg = self.config
        part = self.partitions[charge]
        use_ann = self._uses_ann(charge, mode)
        idx = self._get_ann_index(charge) if use_ann else None
        q = queries.to(self.device).contiguous()
        nq = q.n
        k = self._num_candidates
        stride = pm_stride or q.max_peaks()
        xp = torch if device_out else np
        kw = dict(device=self.device) if device_out else {}
        mk = (lambda shape, dt: torch.empty(shape, dtype=dt, **kw)) if device_out else \
             (lambda shape, dt: np.empty(shape, dt))
        best_row = mk((nq,), xp.int32)
        best_score = mk((nq,), xp.float64)
        n_cand = mk((nq,), xp.int32)
        pm_count = mk((nq,), xp.int32)
        # (the kernel writes every slot: matches first, zeros beyond pm_count)
        pm_pairs = (torch.empty((nq, stride, 2), dtype=torch.int32, **kw) if device_out
                    else np.empty((nq, stride, 2), np.uint32))
        knn = mk((nq, k), xp.int64) if (want_knn and use_ann) else None
        _, min_bound, _ = get_dim(cfg.min_mz, cfg.max_mz, cfg.bin_size)
        P = _lib.AslSearchParams(min_bound, cfg.bin_size, HASH_SEED, k, self._num_probe, charge,
                                 float(tol_val), 0 if tol_mode == 'Da' else 1,
                                 cfg.fragment_mz_tolerance, int(cfg.allow_peak_shifts),
                                 int(use_ann))
        _lib.check(_lib.lib().asl_search_batch(
            part.handle, idx._h if idx is not None else None, C.byref(_lib.peaks_struct(q)),
            C.byref(P), _lib.ptr(best_row), _lib.ptr(best_score), _lib.ptr(n_cand),
            _lib.ptr(pm_count), _lib.ptr(pm_pairs), stride, _lib.ptr(knn)))
        if device_out and use_ann and getattr(self, '_pipeline_on', False):
            self._hold(q, best_row, best_score, n_cand, pm_count, pm_pairs, knn)
        return BatchResult(best_row, best_score, n_cand, pm_count, pm_pairs, knn)

    def _get_library_candidates(self, queries: PackedSpectra, charge: int, mode: str):
        """CSR candidate lists (library rows of the charge partition, ascending) after the
        precursor filter -- and, in open+ann mode, after the ANN filter. Diagnostic
        mirror of spectral_library.py:372-455; ``_search_batch`` never materialises it."""
        tol_val, tol_mode = self._tolerance(mode)
        if charge not in self.partitions:
            return None
        part = self.partitions[charge]
        nq = queries.n
        qp = np.ascontiguousarray(queries.precursor_mz.cpu().numpy(), np.float64)
        off = np.empty(nq + 1, np.int32)
        _lib.check(_lib.lib().asl_window_candidates(part.handle, nq, _lib.ptr(qp), charge,
                                                    float(tol_val), 0 if tol_mode == 'Da' else 1,
                                                    _lib.ptr(off), None))
        rows = np.empty(int(off[-1]), np.int64)
        _lib.check(_lib.lib().asl_window_candidates(part.handle, nq, _lib.ptr(qp), charge,
                                                    float(tol_val), 0 if tol_mode == 'Da' else 1,
                                                    _lib.ptr(off), _lib.ptr(rows)))
        lists = [rows[off[i]:off[i + 1]] for i in range(nq)]
        if self.config.mode == 'ann' and mode == 'open' and charge in self._ann_filenames:
            idx = self._get_ann_index(charge)
            _, I = idx.search(self._encode(queries.to(self.device)), self._num_candidates)
            I = I.cpu().numpy()
            lists = [np.intersect1d(l, I[i][I[i] >= 0]) for i, l in enumerate(lists)]
        return lists

    def search_charge_batches(self, query_spectra: Dict[int, PackedSpectra], mode: str
                              ) -> Iterator[Tuple[int, int, int, np.ndarray, float]]:
        """One cascade level over per-charge query sets, batched like
        ``_search_cascade`` (:301-317). Yields ``(charge, query_index_in_set, library_id,
        peak_matches[n,2], score)`` for every query with at least one candidate."""
        bs = self.config.batch_size
        for charge, qs in query_spectra.items():
            for b0 in range(0, qs.n, bs):
                rows = torch.arange(b0, min(b0 + bs, qs.n))
                res = self._search_batch(qs.select(rows), charge, mode)
                if res is None:
                    continue
                part = self.partitions[charge]
                for i in range(len(rows)):
                    if res.best_row[i] >= 0:
                        yield (charge, b0 + i, part.ids[res.best_row[i]], res.peak_matches(i),
                               float(res.best_score[i]))

    # ------------------------------------------------------------------ cascade driver
    def search(self, query, query_meta: Optional[Dict[int, list]] = None,
               library_meta: Optional[Dict[int, list]] = None, score_ssms=None) -> list:
        """``SpectralLibrary.search(query_filename)`` (spectral_library.py:193-262): identify all
        spectra of a query file. ``query`` is a file name (read by the injected ``query_reader``,
        default the reference's ``read_query_file``), an iterable of query spectrum objects
        (``identifier, precursor_mz, precursor_charge`` -- None: tried at 2 and 3, :213-223 --
        ``mz, intensity``, optional ``retention_time, index``), or the packed form of
        ``search_packed``. Queries are preprocessed in batches on the device, low-quality ones
        dropped (:225-228). Returns the identifications as SSM records with the attributes
        ``writer.write_mztab`` consumes (writer.py:129-148)."""
        if isinstance(query, dict):
            return self.search_packed(query, query_meta, library_meta, score_ssms)
        from .library_store import pack_queries
        if isinstance(query, (str, os.PathLike)):
            logging.info('Process file %s', query)
            query = self._query_reader(os.fspath(query))
        query_spectra, qmeta = pack_queries(query, self.config, self.device)
        lmeta = library_meta if library_meta is not None else self.library_meta
        if lmeta is None:
            raise ValueError('search(): no library metadata (library built from a PackedSpectra); '
                             'pass library_meta=')
        return self.search_packed(query_spectra, qmeta, lmeta,
                                  score_ssms or getattr(self, '_score_ssms', None))

    def search_packed(self, query_spectra: Dict[int, PackedSpectra], query_meta: Dict[int, list],
                      library_meta: Dict[int, list], score_ssms=None) -> 'SSMTable':
        """``SpectralLibrary.search`` (spectral_library.py:193-262) over packed, already
        processed query spectra split by precursor charge (the file parsing and
        ``process_spectrum`` filtering of :207-228 happen before; queries of unknown charge
        are entered once per candidate charge with the same identifier; inside one charge set
        identifiers are unique).

        ``query_meta[charge][i]`` / ``library_meta[charge][row]``: mappings with the reference's
        attribute names (see ``spectrum.ssms_from_batch``). ``score_ssms`` stands for
        ``utils.score_ssms`` (:319-326, mokapot -- out of scope): called as ``score_ssms(ssms,
        mode)`` with a list of SSM records it assigns ``search_engine_score`` / ``q`` and
        returns the SSMs to keep; a callable with the attribute ``columnar = True`` receives the
        ``SSMTable`` itself instead, writes ``table.q`` (and ``table.score``) and may return a
        keep-mask -- no per-SSM Python objects on the search path. Default: cosine as the score,
        q = 0 (everything accepted).

        Returns the identifications (one per query identifier) as an ``SSMTable``: a sequence
        of SSM records materialised on access (``len``, iteration, indexing), ready for
        ``writer.write_mztab``; its columns (``charge, qrow, lib_row, score, q``) are there for
        consumers that do not want 10^5 Python objects."""
        cfg = self.config
        score_ssms = score_ssms or getattr(self, '_score_ssms', None)
        do_cascade_open = (cfg.precursor_tolerance_mass_open is not None and
                           cfg.precursor_tolerance_mode_open is not None)
        uid = _query_uids(query_meta, list(query_spectra))
        remaining = {z: np.arange(q.n, dtype=np.int64) for z, q in query_spectra.items()}
        # cascade level 1: standard search (:238-245); with a second level only the confident
        # identifications are retained
        t1 = self._search_cascade(query_spectra, query_meta, library_meta, remaining, 'std',
                                  score_ssms, uid)
        n_identified = int((t1.q < cfg.fdr).sum())
        # (message parsed by the reference's notebooks: kept verbatim, spectral_library.py:243)
        logging.info('%d spectra identified after the standard search', n_identified)
        if not do_cascade_open:
            return t1
        t1 = t1.take(t1.q < cfg.fdr)
        # cascade level 2: open search on the queries not identified so far (:249-259)
        if uid is None:
            for z in remaining:
                done = np.zeros(len(remaining[z]), bool)
                done[t1.qrow[t1.charge == z]] = True
                remaining[z] = remaining[z][~done]
        else:
            found = t1.uids(uid)
            remaining = {z: rows[~np.isin(uid[z][rows], found)] for z, rows in remaining.items()}
        t2 = self._search_cascade(query_spectra, query_meta, library_meta, remaining, 'open',
                                  score_ssms, uid)
        n_identified += int((t2.q < cfg.fdr).sum())
        logging.info('%d spectra identified after the open search', n_identified)   # :257
        return SSMTable.concat([t1, t2])

    def _search_cascade(self, query_spectra, query_meta, library_meta, rows_by_charge, mode,
                        score_ssms=None, uid=None) -> 'SSMTable':
        """One cascade level (:264-326): batches of ``batch_size`` same-charge queries through
        ``_search_batch``; per query identifier the FIRST match is kept (the reference compares
        ``search_engine_score`` values that are still NaN at this point, :312-316, so a later
        duplicate never replaces an earlier one). Host work is per batch, not per query."""
        import time
        from . import spectrum_similarity
        bs = self.config.batch_size
        t_level = time.perf_counter()
        n_in = sum(len(r) for r in rows_by_charge.values())
        table = SSMTable(query_meta, library_meta)
        # Phase 1 issues every batch of the level; on one GPU the open-search batches go through
        # the two-stream pipeline (front of batch i+1 under the scan of batch i, no host wait
        # between batches). Phase 2, after one synchronisation, scores the winners and files them.
        d = getattr(self, '_dist', None)
        piped = (self.device.type == 'cuda' and (d is None or d.world == 1) and
                 not getattr(self, '_pipeline_on', False) and getattr(self, 'pipeline_cascade', True) and
                 any(self._uses_ann(z, mode) for z in rows_by_charge))
        if piped:
            self.set_pipeline(True)
        pending = []
        try:
            for charge, rows in rows_by_charge.items():
                rows = np.asarray(rows, np.int64)
                qs = query_spectra[charge]
                for b0 in range(0, len(rows), bs):
                    sel = rows[b0:b0 + bs]
                    if len(sel) == 0:
                        continue
                    whole = len(sel) == qs.n and sel[0] == 0 and sel[-1] == qs.n - 1
                    q = (qs if whole else qs.select(torch.as_tensor(sel))).to(self.device)
                    res = self._search_batch(q, charge, mode, device_out=True)
                    if res is not None:
                        pending.append((charge, sel, q, res))
        finally:
            if piped:
                self.set_pipeline(False)         # synchronises first
        # default search-engine score: the cosine over the winner's peak matches -- every batch's
        # kernel is enqueued before the first result is copied back (a copy waits for its kernel)
        cosines = [spectrum_similarity.ssm_cosine(q, self.partitions[charge].spectra, res.best_row,
                                                  res.pm_pairs, res.pm_count)
                   for charge, sel, q, res in pending]
        for (charge, sel, q, res), cos in zip(pending, cosines):
            table.add_batch(charge, sel, _to_np(res.best_row), _to_np(cos), res)
        if uid is not None:
            table = table.first_per_uid(uid)
        acc = getattr(self, 'level_seconds', None)
        if acc is not None:       # bench.py: wall time, queries in, SSMs out of every cascade level
            torch.cuda.synchronize() if self.device.type == 'cuda' else None
            sec, a, b = acc.get(mode, (0.0, 0, 0))
            acc[mode] = (sec + time.perf_counter() - t_level, a + n_in, b + len(table))
        if score_ssms is None:    # no scorer: cosine (spectrum_similarity.py:81-106), accepted
            cascade = self.config.precursor_tolerance_mass_open is not None
            if (cascade or self.config.model is not None) and not getattr(self, '_warned_model', False):
                # the reference gates level 1 by utils.score_ssms (model or the `--model none`
                # FDR filter); without a scorer level 1 keeps EVERY query that had a
                # standard-window candidate, so the open level only sees the rest
                self._warned_model = True
                logging.warning('no score_ssms callable was given (model=%r): the FDR gate '
                                '(utils.score_ssms / mokapot) stays with the caller; every SSM is '
                                'accepted with its cosine as the score and q = 0%s',
                                self.config.model,
                                ', so the cascade hands only the queries without any '
                                'standard-window candidate to the open search' if cascade else '')
            table.q[:] = 0.0
            return table
        if getattr(score_ssms, 'columnar', False):
            keep = score_ssms(table, mode)
            return table if keep is None else table.take(np.asarray(keep))
        objs = table.materialize()
        for i, o in enumerate(objs):
            o._row = i
        kept = list(score_ssms(objs, mode))
        out = table.take(np.asarray([o._row for o in kept], np.int64))
        out.score = np.asarray([o.search_engine_score for o in kept], np.float64)
        out.q = np.asarray([o.q for o in kept], np.float64)
        return out


def _to_np(a):
    return a.detach().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a)


def _query_uids(query_meta, charges) -> Optional[Dict[int, np.ndarray]]:
    """Integer id per query identifier, shared across the charge sets -- or None when there is
    a single set (identifiers are unique inside a set, so nothing can collide)."""
    if len(charges) <= 1:
        return None
    table: Dict = {}
    out = {}
    for z in charges:
        out[z] = np.fromiter((table.setdefault(m['identifier'], len(table)) for m in query_meta[z]),
                             np.int64, len(query_meta[z]))
    return out


class SSMTable:
    """Spectrum-spectrum matches of a cascade level / the identifications of a search, columnar:
    ``charge``, ``qrow`` (row inside ``query_spectra[charge]``), ``lib_row`` (row inside the
    charge partition), ``score`` (search_engine_score), ``q``. Behaves as a sequence of the
    reference's SSM records (``spectrum.SpectrumSpectrumMatch``: the attributes writer.py:129-148
    reads), built on access from the query / library metadata and the peak matches the device
    emitted (kept per batch, fetched from the device when first needed)."""

    def __init__(self, query_meta, library_meta):
        self.query_meta, self.library_meta = query_meta, library_meta
        self.charge = np.zeros(0, np.int32)
        self.qrow = np.zeros(0, np.int64)
        self.lib_row = np.zeros(0, np.int32)
        self.score = np.zeros(0, np.float64)
        self.q = np.zeros(0, np.float64)
        self.batch = np.zeros(0, np.int32)       # index into _batches
        self.pos = np.zeros(0, np.int32)         # row inside that batch
        self._batches: list = []
        self._pending: list = []

    def add_batch(self, charge, qrows, best_row, score, res) -> None:
        hit = np.nonzero(best_row >= 0)[0]       # queries without a candidate: no SSM (:359)
        b = len(self._batches)
        self._batches.append(res)
        self._pending.append((np.full(len(hit), charge, np.int32), np.asarray(qrows, np.int64)[hit],
                              best_row[hit].astype(np.int32), np.asarray(score, np.float64)[hit],
                              np.full(len(hit), b, np.int32), hit.astype(np.int32)))
        self._flush()

    def _flush(self):
        if not self._pending:
            return
        cols = list(zip(*self._pending))
        self._pending = []
        cat = lambda old, new: np.concatenate([old] + list(new))
        self.charge, self.qrow = cat(self.charge, cols[0]), cat(self.qrow, cols[1])
        self.lib_row, self.score = cat(self.lib_row, cols[2]), cat(self.score, cols[3])
        self.batch, self.pos = cat(self.batch, cols[4]), cat(self.pos, cols[5])
        self.q = np.concatenate([self.q, np.full(len(self.charge) - len(self.q), np.nan)])

    def __len__(self):
        return len(self.charge)

    def take(self, sel) -> 'SSMTable':
        """Rows ``sel`` (boolean mask or indices, order kept); shares the per-batch storage."""
        sel = np.asarray(sel)
        idx = np.nonzero(sel)[0] if sel.dtype == bool else sel.astype(np.int64)
        out = SSMTable(self.query_meta, self.library_meta)
        out._batches = self._batches
        for name in ('charge', 'qrow', 'lib_row', 'score', 'q', 'batch', 'pos'):
            setattr(out, name, getattr(self, name)[idx])
        return out

    @staticmethod
    def concat(tables) -> 'SSMTable':
        out = SSMTable(tables[0].query_meta, tables[0].library_meta)
        shift = 0
        for t in tables:
            out._batches = out._batches + t._batches
            for name in ('charge', 'qrow', 'lib_row', 'score', 'q', 'pos'):
                setattr(out, name, np.concatenate([getattr(out, name), getattr(t, name)]))
            out.batch = np.concatenate([out.batch, t.batch + shift])
            shift += len(t._batches)
        return out

    def uids(self, uid) -> np.ndarray:
        out = np.zeros(len(self), np.int64)
        for z in np.unique(self.charge):
            m = self.charge == z
            out[m] = uid[int(z)][self.qrow[m]]
        return out

    def first_per_uid(self, uid) -> 'SSMTable':
        u = self.uids(uid)
        _, first = np.unique(u, return_index=True)
        return self if len(first) == len(u) else self.take(np.sort(first))

    def identifiers(self) -> list:
        return [self.query_meta[int(z)][int(r)]['identifier'] for z, r in zip(self.charge, self.qrow)]

    def library_identifiers(self, partitions) -> np.ndarray:
        """Library identifiers of the matches (``partitions``: ``SpectralLibrary.partitions``)."""
        out = np.empty(len(self), dtype=object)
        for z in np.unique(self.charge):
            m = self.charge == z
            out[m] = partitions[int(z)].ids[self.lib_row[m]]
        return out

    def _column(self, meta, rows, name, dtype, default=None) -> np.ndarray:
        """One metadata attribute of the matches as an array. A metadata container may offer
        ``column(name, rows)`` (arrays behind it) -- otherwise it is read record by record."""
        out = np.zeros(len(self), dtype)
        for z in np.unique(self.charge):
            m = self.charge == z
            cont, r = meta[int(z)], rows[m]
            if hasattr(cont, 'column'):
                out[m] = cont.column(name, r)
            elif default is None:
                out[m] = [cont[int(i)][name] for i in r]
            else:
                out[m] = [cont[int(i)].get(name, default) for i in r]
        return out

    def is_decoy(self) -> np.ndarray:
        return self._column(self.library_meta, self.lib_row, 'is_decoy', bool, False)

    def mass_diffs(self) -> np.ndarray:
        """(experimental - library precursor m/z) * query charge, the quantity the reference
        groups open-search SSMs by (utils.py:227-232)."""
        exp = self._column(self.query_meta, self.qrow, 'precursor_mz', np.float64)
        z = self._column(self.query_meta, self.qrow, 'precursor_charge', np.float64)
        calc = self._column(self.library_meta, self.lib_row, 'precursor_mz', np.float64)
        return (exp - calc) * z

    def _peak_matches(self, i) -> np.ndarray:
        b = self._batches[int(self.batch[i])]
        if not isinstance(b, tuple):             # first access: one device -> host copy per batch
            b = (_to_np(b.pm_count), _to_np(b.pm_pairs))
            self._batches[int(self.batch[i])] = b
        p = int(self.pos[i])
        return b[1][p, :b[0][p]].astype(np.int64)

    def __getitem__(self, i):
        from .spectrum import SpectrumSpectrumMatch
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        z = int(self.charge[i])
        qm, lm = self.query_meta[z][int(self.qrow[i])], self.library_meta[z][int(self.lib_row[i])]
        return SpectrumSpectrumMatch(
            lm['peptide'], qm['identifier'], qm['index'], lm['identifier'],
            qm.get('retention_time'), qm['precursor_charge'], qm['precursor_mz'],
            lm['precursor_mz'], lm.get('is_decoy', False), float(self.score[i]), float(self.q[i]),
            self._peak_matches(i))

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def materialize(self) -> list:
        return list(self)

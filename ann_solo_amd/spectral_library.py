"""Search engine -- host-side mirror of the hot-path half of the reference's
``ann_solo/spectral_library.py`` (``SpectralLibrary`` :27-500).

What is mirrored (same names, same meaning): ``_get_hyperparameter_hash`` (:118-131),
``_create_ann_indexes`` (:133-183), ``_get_ann_index`` (:457-500), ``_search_batch``
(:328-370), ``_get_library_candidates`` (:372-455), ``_search_cascade`` batching
(:301-317) and ``shutdown`` (:185-191). What is different by design: the library is a
packed, HBM-resident peak store per precursor charge instead of per-spectrum HDF5
reads, and a whole batch runs through ``asl_search_batch`` in one device pipeline
(encode -> IVF top-k -> precursor post-filter -> shifted-dot best match).

File parsing, FDR/mokapot scoring and mzTab writing stay with the reference
(out of scope); ``search_charge_batches`` yields exactly the
``(query, library_match, peak_matches)`` triples ``_search_batch`` yields there.
"""
import ctypes as C
import hashlib
import json
import logging
import os
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from . import faiss_compat as faiss
from .packed import PackedSpectra
from .spectrum import get_dim, spectra_to_vectors, HASH_SEED


@dataclass
class Config:
    """The reference's flags that reach the hot path, same names and defaults
    (/root/reference/src/ann_solo/config.py:71-216). ``index``/``pq_m``/``pq_bits`` are
    the additive flags of this implementation."""
    resolution: Optional[int] = None
    min_mz: int = 11
    max_mz: int = 2010
    remove_precursor: bool = False
    remove_precursor_tolerance: float = 0
    min_intensity: float = 0.01
    min_peaks: int = 10
    min_mz_range: float = 250
    max_peaks_used: int = 50
    max_peaks_used_library: int = 50
    scaling: Optional[str] = 'rank'
    fdr: float = 0.01
    fdr_min_group_size: int = 100
    spectral_library_filename: str = ''
    query_filename: str = ''
    bin_size: float = 0.04
    hash_len: int = 800
    num_candidates: int = 1024
    batch_size: int = 16384
    num_list: int = 256
    num_probe: int = 128
    mode: str = 'ann'                       # 'ann' | 'bf'
    precursor_tolerance_mass: float = 20.0
    precursor_tolerance_mode: str = 'ppm'   # 'Da' | 'ppm'
    precursor_tolerance_mass_open: Optional[float] = 300.0
    precursor_tolerance_mode_open: Optional[str] = 'Da'
    fragment_mz_tolerance: float = 0.02
    allow_peak_shifts: bool = True
    no_gpu: bool = False
    index: str = 'ivfflat'                  # 'ivfflat' | 'ivfpq'
    pq_m: int = 32
    pq_bits: int = 8
    kmeans_niter: int = 25
    seed: int = 1234

    def __getitem__(self, k):
        return getattr(self, k)


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


class SpectralLibrary:
    _hyperparameters = ['min_mz', 'max_mz', 'bin_size', 'hash_len', 'num_list']

    def __init__(self, library: PackedSpectra, identifiers=None, config: Config = None,
                 valid: Optional[np.ndarray] = None, index_dir: Optional[str] = None,
                 basename: str = 'library', device='cuda'):
        """``library``: all library spectra, already processed (``process_spectrum``);
        ``identifiers``: library ids (default: row numbers); ``valid``: is_valid flags.
        ANN indexes are built for every charge with >= num_list spectra
        (spectral_library.py:100-104) and cached as ``<base>_<hash7>_<charge>.idxann``
        under ``index_dir`` when given (:98-108)."""
        self.config = config or Config()
        self.device = torch.device(device)
        cfg = self.config
        if cfg.num_candidates > 2048 or cfg.num_probe > 2048:
            # FAISS-GPU clamps both to 1024 (:76-87); this implementation's LDS top-k holds 2048
            logging.warning('Using num_candidates/num_probe <= 2048 (maximum supported)')
            cfg.num_candidates = min(cfg.num_candidates, 2048)
            cfg.num_probe = min(cfg.num_probe, 2048)
        self._num_probe = cfg.num_probe
        self._num_candidates = cfg.num_candidates
        self._use_gpu = True
        self._ann_filenames: Dict[int, str] = {}
        self._current_index: Tuple[Optional[int], Optional[faiss.Index]] = (None, None)
        self._index_dir = index_dir
        self._basename = basename
        self.partitions: Dict[int, ChargePartition] = {}
        n = library.n
        ids = np.arange(n) if identifiers is None else np.asarray(identifiers)
        pz = library.precursor_charge.cpu().numpy()
        valid = np.ones(n, bool) if valid is None else np.asarray(valid, bool)
        for z in np.unique(pz):
            rows = np.nonzero(pz == z)[0]
            part = library.select(torch.as_tensor(rows)).to(self.device).contiguous()
            pmz32 = part.precursor_mz.cpu().numpy().astype(np.float32)
            v = np.ascontiguousarray(valid[rows].astype(np.uint8))
            h = _lib.lib().asl_library_create(C.byref(_lib.peaks_struct(part)), _lib.ptr(pmz32),
                                              _lib.ptr(v))
            if not h:
                _lib.check(-1)
            self.partitions[int(z)] = ChargePartition(int(z), ids[rows], pmz32, part,
                                                      C.c_void_p(h))
        if cfg.mode == 'ann':
            create = []
            for z in sorted(self.partitions):
                if len(self.partitions[z].ids) < cfg.num_list:
                    continue          # infrequent charge: brute force (spectral_library.py:102-104)
                base = f'{self._basename}_{self._get_hyperparameter_hash()[:7]}'
                self._ann_filenames[z] = os.path.join(index_dir or '', f'{base}_{z}.idxann')
                if index_dir is None or not os.path.isfile(self._ann_filenames[z]):
                    create.append(z)
            if create:
                self._create_ann_indexes(create)

    # ------------------------------------------------------------------ reference mirrors
    def _get_hyperparameter_hash(self) -> str:
        b = json.dumps({hp: self.config[hp] for hp in self._hyperparameters}).encode('utf-8')
        return hashlib.sha1(b).hexdigest()

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
                                               faiss.METRIC_INNER_PRODUCT)
            ann_index.seed = cfg.seed
            ann_index.set_niter(cfg.kmeans_niter)
            ann_index.train(vectors)
            ann_index.add(vectors)
            if self._index_dir is not None:
                faiss.write_index(ann_index, self._ann_filenames[z])
            part.index = ann_index
            del vectors

    def _get_ann_index(self, charge: int) -> faiss.Index:
        part = self.partitions[charge]
        if part.index is None:
            part.index = faiss.read_index(self._ann_filenames[charge])
        part.index.nprobe = self._num_probe
        self._current_index = charge, part.index
        return part.index

    def shutdown(self) -> None:
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

    def _search_batch(self, queries: PackedSpectra, charge: int, mode: str,
                      want_knn: bool = False, device_out: bool = False) -> Optional[BatchResult]:
        """One batch of same-charge, processed query spectra through the device hot path.
        Returns None when the library has no spectra of that charge (:411-412)."""
        tol_val, tol_mode = self._tolerance(mode)
        if tol_mode not in ('Da', 'ppm'):
            raise ValueError('Unknown precursor tolerance mode')
        if charge not in self.partitions:
            return None
        cfg = self.config
        part = self.partitions[charge]
        use_ann = cfg.mode == 'ann' and mode == 'open' and charge in self._ann_filenames
        idx = self._get_ann_index(charge) if use_ann else None
        q = queries.to(self.device).contiguous()
        nq = q.n
        k = self._num_candidates
        stride = q.max_peaks()
        xp = torch if device_out else np
        kw = dict(device=self.device) if device_out else {}
        mk = (lambda shape, dt: torch.empty(shape, dtype=dt, **kw)) if device_out else \
             (lambda shape, dt: np.empty(shape, dt))
        best_row = mk((nq,), xp.int32)
        best_score = mk((nq,), xp.float64)
        n_cand = mk((nq,), xp.int32)
        pm_count = mk((nq,), xp.int32)
        pm_pairs = (torch.zeros((nq, stride, 2), dtype=torch.int32, **kw) if device_out
                    else np.zeros((nq, stride, 2), np.uint32))
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
    def search(self, query_spectra: Dict[int, PackedSpectra], query_meta: Dict[int, list],
               library_meta: Dict[int, list], score_ssms=None) -> list:
        """``SpectralLibrary.search`` (spectral_library.py:193-262) over packed, already
        processed query spectra split by precursor charge (the file parsing and
        ``process_spectrum`` filtering of :207-228 happen before; queries of unknown charge
        are entered once per candidate charge with the same identifier).

        ``query_meta[charge][i]`` / ``library_meta[charge][row]``: mappings with the reference's
        attribute names (see ``writer.ssms_from_batch``). ``score_ssms(ssms, mode)`` stands
        for ``utils.score_ssms`` (:319-326, mokapot -- out of scope): it assigns
        ``search_engine_score`` / ``q`` and returns the SSMs to keep; default: cosine as the
        score, q = 0 (everything accepted). Returns the list of identifications (one per
        query identifier), ready for ``writer.write_mztab``."""
        cfg = self.config
        identifications = {}
        do_cascade_open = (cfg.precursor_tolerance_mass_open is not None and
                           cfg.precursor_tolerance_mode_open is not None)
        remaining = {z: list(range(q.n)) for z, q in query_spectra.items()}
        # cascade level 1: standard search (:238-245)
        for ssm in self._search_cascade(query_spectra, query_meta, library_meta, remaining, 'std',
                                        score_ssms):
            if not do_cascade_open or ssm.q < cfg.fdr:
                identifications[ssm.query_identifier] = ssm
        if do_cascade_open:
            # cascade level 2: open search on the queries not identified so far (:249-259)
            remaining = {z: [i for i in rows if query_meta[z][i]['identifier'] not in identifications]
                         for z, rows in remaining.items()}
            for ssm in self._search_cascade(query_spectra, query_meta, library_meta, remaining,
                                            'open', score_ssms):
                identifications[ssm.query_identifier] = ssm
        return list(identifications.values())

    def _search_cascade(self, query_spectra, query_meta, library_meta, rows_by_charge, mode,
                        score_ssms=None) -> list:
        """One cascade level (:264-326): batches of ``batch_size`` same-charge queries through
        ``_search_batch``; per query identifier the FIRST match is kept (the reference compares
        ``search_engine_score`` values that are still NaN at this point, :312-316, so a later
        duplicate never replaces an earlier one)."""
        from .spectrum_similarity import ssm_features
        from .writer import ssms_from_batch
        ssms = {}
        bs = self.config.batch_size
        for charge, rows in rows_by_charge.items():
            for b0 in range(0, len(rows), bs):
                sel = rows[b0:b0 + bs]
                if not sel:
                    continue
                q = query_spectra[charge].select(torch.as_tensor(sel, dtype=torch.int64))
                res = self._search_batch(q, charge, mode)
                if res is None:
                    continue
                part = self.partitions[charge]
                cos = ssm_features(q.to(self.device), part.spectra, res.best_row, res.pm_pairs,
                                   res.pm_count, self.config.min_mz, self.config.max_mz,
                                   self.config.bin_size)[:, 0]
                cos = cos.detach().cpu().numpy() if hasattr(cos, 'detach') else cos
                qm = [query_meta[charge][i] for i in sel]
                for ssm in ssms_from_batch(res, qm, library_meta[charge], scores=cos):
                    if ssm.query_identifier not in ssms:
                        ssms[ssm.query_identifier] = ssm
        out = list(ssms.values())
        if score_ssms is not None:
            return list(score_ssms(out, mode))
        for ssm in out:        # no scorer: cosine (spectrum_similarity.py:81-106), accepted
            ssm.q = 0.0
        return out


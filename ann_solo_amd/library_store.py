"""Adapter between the reference's readers and the packed, device-resident peak store
(SURVEY.md 8 rows b4, f1).

The reference keeps library spectra in an HDF5 store and materialises + preprocesses every
candidate on demand (``SpectralLibraryReader.read_spectrum(id, True)``,
/root/reference/src/ann_solo/reader.py:218-246; ``spectral_library.py:449-455``). Here the
whole library is read ONCE through the reader's public surface --

    reader.spec_info['charge'][z] = {'id': ndarray, 'precursor_mz': float32 ndarray}
                                                            (reader.py:180-191)
    reader.read_all_spectra()  -> iterator of spectrum objects (reader.py:249-259)
    reader.is_recreated        -> the cached files were rebuilt (reader.py:161)
    reader.get_version()       -> database version for the mzTab writer (reader.py:289-298)

-- preprocessed in batches by the HIP ``process_spectrum`` kernel and written as one packed
store ``<library>_<hash7>.spstore`` next to the reference's ``.spcfg``/``.hdf5`` pair. Row r
of a charge partition IS ``spec_info['charge'][z]['id'][r]`` by construction (the reference
adds index rows in HDF5 key order but masks columns in ``spec_info`` order -- SURVEY.md 9.2).

A spectrum object needs ``identifier, precursor_mz, precursor_charge, mz, intensity`` and
optionally ``annotation`` (objects with ``.charge`` or None) / ``charge``, ``peptide``,
``is_decoy``, ``retention_time``, ``index`` -- the attributes of the reference's
``MsmsSpectrum`` subclass that the hot path and the writer consume.

No file parsing happens here: ``.splib/.sptxt/.mgf`` parsing stays with the reference's reader
(out of scope).
"""
import hashlib
import json
import logging
import os
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from .packed import PackedSpectra

STORE_EXT = '.spstore'
# preprocessing options that change the processed peaks of a LIBRARY spectrum
# (/root/reference/src/ann_solo/spectrum.py:57-119 with is_library=True)
_PROCESS_KEYS = ['resolution', 'min_mz', 'max_mz', 'remove_precursor',
                 'remove_precursor_tolerance', 'min_intensity', 'min_peaks', 'min_mz_range',
                 'max_peaks_used_library', 'scaling']


def store_hash(config, hyperparameter_hash: str, annotation_alignment: str) -> str:
    """Key of a packed store: the reference's hyper-parameter hash (what its ``.spcfg`` is keyed
    by) + every option that changes the processed library peaks + the store layout version."""
    from .config import Config
    config = Config.from_reference(config)
    opts = {k: getattr(config, k, None) for k in _PROCESS_KEYS}
    b = json.dumps({'hyper': hyperparameter_hash, 'process': opts,
                    'annotation': annotation_alignment, 'layout': 2}, sort_keys=True)
    return hashlib.sha1(b.encode('utf-8')).hexdigest()


def _sorted_peaks(s):
    mz = np.asarray(s.mz, np.float32)
    it = np.asarray(s.intensity, np.float32)
    chg = getattr(s, 'charge', None)
    if chg is None:
        ann = getattr(s, 'annotation', None)
        chg = np.zeros(len(mz), np.uint8)
        if ann is not None:                       # spectrum_match.pyx:74-79
            for i, a in enumerate(ann):
                if i < len(chg) and a is not None:
                    chg[i] = getattr(a, 'charge', 0) or 0
    chg = np.asarray(chg, np.uint8)
    if len(chg) != len(mz):                       # defensive: treat as unannotated
        chg = np.zeros(len(mz), np.uint8)
    if len(mz) > 1 and (np.diff(mz) < 0).any():   # MsmsSpectrum sorts on construction; be sure
        o = np.argsort(mz, kind='stable')
        mz, it, chg = mz[o], it[o], chg[o]
    return mz, it, chg


def pack_raw(spectra: Iterable, charges: Optional[List[int]] = None) -> PackedSpectra:
    """Raw (unprocessed) spectrum objects -> PackedSpectra on the host; ``charges`` overrides the
    objects' precursor charges (queries of unknown charge are tried at 2 and 3)."""
    offs, mzs, its, chgs, pmz, pz, ids = [0], [], [], [], [], [], []
    for i, s in enumerate(spectra):
        mz, it, chg = _sorted_peaks(s)
        mzs.append(mz)
        its.append(it)
        chgs.append(chg)
        offs.append(offs[-1] + len(mz))
        pmz.append(float(s.precursor_mz))
        pz.append(int(charges[i] if charges is not None else s.precursor_charge))
        ids.append(getattr(s, 'identifier', None))
    cat = (lambda xs, dt: np.concatenate(xs).astype(dt, copy=False) if xs else np.zeros(0, dt))
    return PackedSpectra.from_numpy(np.asarray(offs), cat(mzs, np.float32), cat(its, np.float32),
                                    cat(chgs, np.uint8), np.asarray(pmz, np.float64),
                                    np.asarray(pz, np.int32), 'cpu', ids)


def _snapshot_annotation(raw: PackedSpectra, out: PackedSpectra) -> PackedSpectra:
    """The reference snapshot restores the RAW annotation array on a processed library spectrum
    (``spectrum._annotation = annotation``, reader.py:243-245), so peak j of the processed
    spectrum carries the annotation of RAW peak j (spectrum_match.pyx:74-85 reads ``charge[j]``
    for j < len(mz)). Opt-in reproduction of that alignment."""
    ro = raw.offsets.to(torch.int64)
    oo = out.offsets.to(torch.int64)
    cnt = oo[1:] - oo[:-1]
    seg = torch.repeat_interleave(torch.arange(out.n, device=oo.device), cnt)
    pos = torch.arange(int(oo[-1]), device=oo.device) - oo[seg] + ro[seg]
    return PackedSpectra(out.offsets, out.mz, out.intensity, raw.charge.to(out.charge.device)[pos],
                         out.precursor_mz, out.precursor_charge, identifiers=out.identifiers)


def process_in_chunks(items: Iterable, is_library: bool, config, device, chunk: int = 32768,
                      annotation_alignment: str = 'peaks'):
    """Batched ``process_spectrum`` over an iterator of ``(spectrum object, precursor charge,
    tag)`` triples: yields ``(objects, charges, tags, processed PackedSpectra on the host,
    valid bool ndarray)`` per chunk."""
    from .spectrum import process_spectra
    buf, chg, tags = [], [], []

    def flush():
        raw = pack_raw(buf, chg)
        out, valid = process_spectra(raw, is_library, config, device)
        if annotation_alignment == 'snapshot':
            out = _snapshot_annotation(raw.to(out.device), out)
        return list(buf), list(chg), list(tags), out.to('cpu'), valid.cpu().numpy()
    for s, z, tag in items:
        buf.append(s)
        chg.append(int(z))
        tags.append(tag)
        if len(buf) >= chunk:
            yield flush()
            buf, chg, tags = [], [], []
    if buf:
        yield flush()


@dataclass
class LibraryStore:
    """All library spectra, processed, charge-major in ``spec_info`` order."""
    spectra: PackedSpectra                  # host
    valid: np.ndarray                       # bool [n]: is_valid after process_spectrum
    ranges: Dict[int, Tuple[int, int]]      # charge -> [first row, last row) of ``spectra``
    meta: Dict[int, List[dict]]             # charge -> per-row identifier / peptide / precursor_mz / is_decoy


def build_library_store(reader, config, device, annotation_alignment: str = 'peaks',
                        chunk: int = 32768) -> LibraryStore:
    spec_info = reader.spec_info['charge']
    ranges, row_of, base = {}, {}, 0
    for z, info in spec_info.items():
        ids = list(np.asarray(info['id']).tolist())
        ranges[int(z)] = (base, base + len(ids))
        for r, i in enumerate(ids):
            row_of[(int(z), i)] = base + r
        base += len(ids)
    n = base
    counts = np.zeros(n, np.int64)
    valid = np.zeros(n, bool)
    pmz = np.zeros(n, np.float64)
    pz = np.zeros(n, np.int32)
    meta_flat: List[Optional[dict]] = [None] * n
    pieces = []                                # (rows ndarray, processed pack) per chunk
    seen = 0
    items = ((s, s.precursor_charge, None) for s in reader.read_all_spectra())
    for objs, chgs, _, out, ok in process_in_chunks(items, True, config, device, chunk,
                                                    annotation_alignment):
        rows = np.full(len(objs), -1, np.int64)
        for j, (s, z) in enumerate(zip(objs, chgs)):
            r = row_of.get((z, s.identifier))
            if r is None:
                continue                       # not listed in spec_info: the reference never sees it
            rows[j] = r
            valid[r] = bool(ok[j])
            pmz[r] = float(s.precursor_mz)
            pz[r] = z
            meta_flat[r] = dict(identifier=s.identifier, peptide=getattr(s, 'peptide', None),
                                precursor_mz=float(s.precursor_mz),
                                is_decoy=bool(getattr(s, 'is_decoy', False)))
            seen += 1
        o = out.offsets.numpy().astype(np.int64)
        counts[rows[rows >= 0]] = (o[1:] - o[:-1])[rows >= 0]
        pieces.append((rows, out))
    if seen != n:
        missing = n - seen
        raise ValueError(f'{missing} spectra of spec_info were not returned by read_all_spectra()')
    offsets = np.zeros(n + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    P = int(offsets[-1])
    mz = np.zeros(P, np.float32)
    it = np.zeros(P, np.float32)
    chg = np.zeros(P, np.uint8)
    for rows, out in pieces:
        o = out.offsets.numpy().astype(np.int64)
        omz, oit, ochg = out.mz.numpy(), out.intensity.numpy(), out.charge.numpy()
        sel = np.nonzero(rows >= 0)[0]
        cnt = (o[1:] - o[:-1])[sel]
        within = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        src = np.repeat(o[:-1][sel], cnt) + within
        dst = np.repeat(offsets[rows[sel]], cnt) + within
        mz[dst], it[dst], chg[dst] = omz[src], oit[src], ochg[src]
    ids_flat = [m['identifier'] for m in meta_flat]
    pack = PackedSpectra.from_numpy(offsets, mz, it, chg, pmz, pz, 'cpu', ids_flat)
    meta = {z: meta_flat[a:b] for z, (a, b) in ranges.items()}
    return LibraryStore(pack, valid, ranges, meta)


def save_library_store(store: LibraryStore, path: str, key: str) -> None:
    extra = {'valid': np.packbits(store.valid).tobytes().hex(),
             'ranges': {str(z): list(r) for z, r in store.ranges.items()},
             'peptide': [m['peptide'] for z in store.ranges for m in store.meta[z]],
             'is_decoy': [int(m['is_decoy']) for z in store.ranges for m in store.meta[z]]}
    store.spectra.save(path, key, extra=extra)


def load_library_store(path: str, key: str) -> LibraryStore:
    pack, meta = PackedSpectra.load(path, key, return_meta=True)
    extra = meta.get('extra') or {}
    n = pack.n
    valid = np.unpackbits(np.frombuffer(bytes.fromhex(extra['valid']), np.uint8))[:n].astype(bool)
    ranges = {int(z): tuple(r) for z, r in extra['ranges'].items()}
    pmz = pack.precursor_mz.numpy()
    flat = [dict(identifier=pack.identifiers[i], peptide=extra['peptide'][i],
                 precursor_mz=float(pmz[i]), is_decoy=bool(extra['is_decoy'][i]))
            for i in range(n)]
    return LibraryStore(pack, valid, ranges, {z: flat[a:b] for z, (a, b) in ranges.items()})


def load_or_build_library_store(reader, config, device, path: Optional[str], key: str,
                                annotation_alignment: str = 'peaks') -> LibraryStore:
    """The packed store of ``reader``'s library: read from ``path`` when it was written under
    the same key and the reader did not just recreate its own caches, else built (and written
    when ``path`` is given). A store whose identifiers disagree with ``spec_info`` is rebuilt."""
    if path and os.path.isfile(path) and not getattr(reader, 'is_recreated', False):
        try:
            st = load_library_store(path, key)
            si = reader.spec_info['charge']
            same = set(st.ranges) == {int(z) for z in si} and all(
                list(np.asarray(si[z]['id']).tolist()) ==
                [m['identifier'] for m in st.meta[int(z)]] for z in si)
            if same:
                return st
            logging.warning('Packed library store %s does not match the library: rebuilding', path)
        except (ValueError, KeyError, OSError) as e:
            logging.warning('Packed library store %s unusable (%s): rebuilding', path, e)
    st = build_library_store(reader, config, device, annotation_alignment)
    if path:
        save_library_store(st, path, key)
    return st


def pack_queries(spectra: Iterable, config, device, chunk: int = 32768):
    """Query side of ``SpectralLibrary.search`` (spectral_library.py:207-228): queries of unknown
    precursor charge are tried at 2 and 3, every copy is preprocessed (HIP ``process_spectrum``)
    and low-quality ones are dropped. Returns ``({charge: PackedSpectra}, {charge: [meta]})``
    with charges in order of first appearance and spectra in file order, as the reference's
    ``defaultdict(list)`` holds them."""
    per_charge: Dict[int, list] = {}
    metas: Dict[int, list] = {}

    def items():
        for pos, s in enumerate(spectra):
            for z in ([s.precursor_charge] if s.precursor_charge is not None else [2, 3]):
                yield s, z, pos
    for objs, chgs, tags, out, ok in process_in_chunks(items(), False, config, device, chunk):
        rows_by_z: Dict[int, list] = {}
        for j, (s, z, pos) in enumerate(zip(objs, chgs, tags)):
            if not ok[j]:
                continue
            rows_by_z.setdefault(z, []).append(j)
            metas.setdefault(z, []).append(dict(
                identifier=s.identifier, index=getattr(s, 'index', pos),
                retention_time=getattr(s, 'retention_time', None), precursor_charge=z,
                precursor_mz=float(s.precursor_mz)))
        for z, rows in rows_by_z.items():
            per_charge.setdefault(z, []).append(out.select(torch.as_tensor(rows)))
    return {z: concat_packs(per_charge[z]) for z in metas}, metas


def concat_packs(packs: List[PackedSpectra]) -> PackedSpectra:
    if len(packs) == 1:
        return packs[0]
    offs = [packs[0].offsets.to(torch.int64)]
    base = int(offs[0][-1])
    for p in packs[1:]:
        offs.append(p.offsets.to(torch.int64)[1:] + base)
        base += int(p.offsets[-1])
    ids = None
    if all(p.identifiers is not None for p in packs):
        ids = [i for p in packs for i in p.identifiers]
    return PackedSpectra(torch.cat(offs).to(torch.int32), torch.cat([p.mz for p in packs]),
                         torch.cat([p.intensity for p in packs]),
                         torch.cat([p.charge for p in packs]),
                         torch.cat([p.precursor_mz for p in packs]),
                         torch.cat([p.precursor_charge for p in packs]), identifiers=ids)

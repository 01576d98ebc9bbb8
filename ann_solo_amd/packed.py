"""Packed (SoA) spectra: the device-resident replacement of the per-spectrum
``MsmsSpectrum`` objects the reference materialises one HDF5 read at a time
(/root/reference/src/ann_solo/reader.py:218-246, spectral_library.py:449-455).

Layout: ``offsets i32[n+1]``, ``mz f32[]``, ``intensity f32[]``,
``charge u8[]`` (fragment-charge annotation, 0 = none; what spectrum_match.pyx:74-79
derives from ``annotation``), ``precursor_mz f64[n]``, ``precursor_charge i32[n]``.
Peaks of one spectrum are ascending in m/z and already processed
(``process_spectrum``, spectrum.py:57-119).
"""
import json
import os
import tempfile
import struct
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

STORE_MAGIC = b'ASLPKS01'


@dataclass
class PackedSpectra:
    offsets: torch.Tensor            # int32 [n+1]
    mz: torch.Tensor                 # float32 [P]
    intensity: torch.Tensor          # float32 [P]
    charge: torch.Tensor             # uint8 [P]
    precursor_mz: torch.Tensor       # float64 [n]
    precursor_charge: torch.Tensor   # int32 [n]
    identifiers: Optional[list] = None

    @property
    def n(self) -> int:
        return self.offsets.numel() - 1

    def __len__(self):
        return self.n

    @property
    def device(self):
        return self.mz.device

    def max_peaks(self) -> int:
        """Largest peak count of a spectrum (>= 1); computed once per pack and cached -- it
        sizes the peak-match tables and would otherwise cost a device round trip per batch."""
        m = getattr(self, '_max_peaks', None)
        if m is None:
            m = int((self.offsets[1:] - self.offsets[:-1]).max()) if self.n else 1
            self._max_peaks = m = max(m, 1)
        return m

    def to(self, device) -> 'PackedSpectra':
        if torch.device(device) == self.mz.device:
            return self
        out = PackedSpectra(*(t.to(device) for t in self._tensors()),
                            identifiers=self.identifiers)
        if getattr(self, '_max_peaks', None) is not None:
            out._max_peaks = self._max_peaks
        return out

    def _tensors(self):
        return (self.offsets, self.mz, self.intensity, self.charge, self.precursor_mz,
                self.precursor_charge)

    def contiguous(self):
        if all(t.is_contiguous() for t in self._tensors()):
            return self
        out = PackedSpectra(*(t.contiguous() for t in self._tensors()),
                            identifiers=self.identifiers)
        if getattr(self, '_max_peaks', None) is not None:
            out._max_peaks = self._max_peaks
        return out

    def select(self, rows) -> 'PackedSpectra':
        """Gather a subset of spectra (rows: 1-D int tensor/array) into a new pack."""
        rows = torch.as_tensor(rows, dtype=torch.int64, device=self.device)
        off = self.offsets.to(torch.int64)
        cnt = off[rows + 1] - off[rows]
        new_off = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=self.device)
        new_off[1:] = torch.cumsum(cnt, 0)
        total = int(new_off[-1])
        seg = torch.repeat_interleave(torch.arange(rows.numel(), device=self.device), cnt,
                                      output_size=total)
        pos = torch.arange(total, device=self.device) - new_off[seg] + off[rows][seg]
        ids = None
        if self.identifiers is not None:
            ids = [self.identifiers[int(r)] for r in rows.cpu()]
        return PackedSpectra(new_off.to(torch.int32), self.mz[pos], self.intensity[pos],
                             self.charge[pos], self.precursor_mz[rows],
                             self.precursor_charge[rows], identifiers=ids)

    def numpy(self):
        """(offsets, mz, intensity, charge, precursor_mz, precursor_charge) as numpy."""
        return tuple(t.detach().cpu().numpy() for t in self._tensors())

    @staticmethod
    def from_numpy(offsets, mz, intensity, charge, precursor_mz, precursor_charge,
                   device='cpu', identifiers=None) -> 'PackedSpectra':
        if charge is None:
            charge = np.zeros(len(mz), np.uint8)
        return PackedSpectra(
            torch.as_tensor(np.ascontiguousarray(offsets, np.int32), device=device),
            torch.as_tensor(np.ascontiguousarray(mz, np.float32), device=device),
            torch.as_tensor(np.ascontiguousarray(intensity, np.float32), device=device),
            torch.as_tensor(np.ascontiguousarray(charge, np.uint8), device=device),
            torch.as_tensor(np.ascontiguousarray(precursor_mz, np.float64), device=device),
            torch.as_tensor(np.ascontiguousarray(precursor_charge, np.int32), device=device),
            identifiers=identifiers)

    # ------------------------------------------------------------------ on-disk store
    # the processed-peak store that replaces the per-candidate HDF5 reads
    # of reader.py:218-246,523-556. One little-endian file:
    #   magic 'ASLPKS01' | u64 n | u64 P | u32 len(meta) | meta (utf-8 JSON: hyper-hash,
    #   identifiers) | offsets i64[n+1] | mz f32[P] | intensity f32[P] | charge u8[P] |
    #   precursor_mz f64[n] | precursor_charge u8[n]
    # keyed by the SHA-1 hyper-parameter hash the reference keeps in its .spcfg file
    # (reader.py: config hash check) -- a store written under other settings is rejected.
    def save(self, path: str, hyperparameter_hash: str = '', extra: Optional[dict] = None) -> None:
        o, mz, it, chg, pmz, pz = self.numpy()
        if pz.size and (pz.min() < 0 or pz.max() > 255):
            raise ValueError('precursor charge outside 0..255')
        ids = None if self.identifiers is None else [
            i if isinstance(i, (str, type(None))) else int(i) for i in self.identifiers]
        meta = {'hash': hyperparameter_hash, 'identifiers': ids}
        if extra is not None:
            meta['extra'] = extra
        meta = json.dumps(meta).encode('utf-8')
        # written under a private name and renamed: the ranks of a sharded job build the same
        # store at the same time, and a reader must never see a half-written file
        fd, tmp = tempfile.mkstemp(dir=os.path.dirname(os.path.abspath(path)) or '.',
                                   prefix=os.path.basename(path) + '.tmp')   # unique across hosts sharing the directory
        os.close(fd)
        try:
            with open(tmp, 'wb') as f:
                f.write(STORE_MAGIC)
                f.write(struct.pack('<QQI', self.n, int(mz.shape[0]), len(meta)))
                f.write(meta)
                for a, dt in ((o, '<i8'), (mz, '<f4'), (it, '<f4'), (chg, 'u1'), (pmz, '<f8'),
                              (pz, 'u1')):
                    f.write(np.ascontiguousarray(a).astype(dt, copy=False).tobytes())
            os.chmod(tmp, 0o644)     # mkstemp creates 0600
            os.replace(tmp, path)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    @staticmethod
    def load(path: str, hyperparameter_hash: Optional[str] = None,
             device='cpu', return_meta: bool = False) -> 'PackedSpectra':
        """Read a store; if ``hyperparameter_hash`` is given it must match the one the store
        was written under (``ValueError`` otherwise, the reference's ``is_recreated`` case)."""
        with open(path, 'rb') as f:
            if f.read(8) != STORE_MAGIC:
                raise ValueError(f'{path}: not a packed spectrum store')
            n, P, lm = struct.unpack('<QQI', f.read(20))
            meta = json.loads(f.read(lm).decode('utf-8'))
            if hyperparameter_hash is not None and meta.get('hash') != hyperparameter_hash:
                raise ValueError(f'{path}: written under different hyper-parameters')

            def rd(cnt, dt):
                a = np.frombuffer(bytearray(f.read(cnt * np.dtype(dt).itemsize)), dtype=dt)
                if a.shape[0] != cnt:
                    raise ValueError(f'{path}: truncated')
                return a
            o, mz, it = rd(n + 1, '<i8'), rd(P, '<f4'), rd(P, '<f4')
            chg, pmz, pz = rd(P, 'u1'), rd(n, '<f8'), rd(n, 'u1')
        if n and (o[0] != 0 or o[-1] != P or (np.diff(o) < 0).any()):
            raise ValueError(f'{path}: corrupt offsets')
        pack = PackedSpectra.from_numpy(o, mz, it, chg, pmz, pz, device, meta.get('identifiers'))
        return (pack, meta) if return_meta else pack

    @staticmethod
    def from_spectra(spectra, device='cpu') -> 'PackedSpectra':
        """Pack reference-style spectrum objects (``.mz .intensity .precursor_mz
        .precursor_charge`` and either ``.charge`` or ``.annotation``)."""
        offs = [0]
        mzs, ints, chgs, pmz, pz, ids = [], [], [], [], [], []
        for s in spectra:
            mz = np.asarray(s.mz, np.float32)
            mzs.append(mz)
            ints.append(np.asarray(s.intensity, np.float32))
            chg = getattr(s, 'charge', None)
            if chg is None:
                ann = getattr(s, 'annotation', None)
                chg = np.zeros(len(mz), np.uint8)
                if ann is not None:          # spectrum_match.pyx:74-79
                    for i, a in enumerate(ann):
                        if a is not None:
                            chg[i] = a.charge
            chgs.append(np.asarray(chg, np.uint8))
            offs.append(offs[-1] + len(mz))
            pmz.append(float(s.precursor_mz))
            pz.append(int(s.precursor_charge))
            ids.append(getattr(s, 'identifier', None))
        cat = (lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt))
        return PackedSpectra.from_numpy(np.asarray(offs), cat(mzs, np.float32),
                                        cat(ints, np.float32), cat(chgs, np.uint8),
                                        np.asarray(pmz), np.asarray(pz), device, ids)

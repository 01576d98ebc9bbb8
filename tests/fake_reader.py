"""A stand-in with the surface of the reference's ``SpectralLibraryReader`` / query reader
(/root/reference/src/ann_solo/reader.py:40-259) for tests: spectra are plain objects with the
attributes of the reference's ``MsmsSpectrum`` subclass that the hot path and the writer read."""
import numpy as np


class Annotation:
    def __init__(self, charge):
        self.charge = charge


class FakeSpectrum:
    def __init__(self, identifier, precursor_mz, precursor_charge, mz, intensity, annotation=None,
                 peptide=None, is_decoy=False, retention_time=None, index=None):
        self.identifier = identifier
        self.precursor_mz = precursor_mz
        self.precursor_charge = precursor_charge
        self.mz = np.asarray(mz, np.float32)
        self.intensity = np.asarray(intensity, np.float32)
        self.annotation = annotation
        self.peptide = peptide
        self.is_decoy = is_decoy
        self.retention_time = retention_time
        if index is not None:
            self.index = index


def raw_spectrum(rng, identifier, charge, n_peaks=None, peptide=None, good=True):
    n = int(n_peaks if n_peaks is not None else rng.integers(60, 160))
    if good:
        mz = np.sort(rng.uniform(100, 1800, n))
    else:                                          # too narrow an m/z range: invalid after processing
        mz = np.sort(rng.uniform(500, 560, n))
    it = rng.lognormal(0, 1.2, n)
    ann = [Annotation(int(rng.integers(1, 3))) if rng.random() < 0.7 else None for _ in range(n)]
    return FakeSpectrum(identifier, float(rng.uniform(350, 1100)), charge, mz, it, ann,
                        peptide or f'PEPTIDE{identifier}K', bool(rng.random() < 0.1))


class FakeReader:
    """``spec_info`` in file order, ``read_all_spectra()`` in HDF5 key order (alphabetical by the
    identifier's string -- the misalignment of SURVEY.md 9.2)."""

    def __init__(self, spectra, filename='/data/lib.splib'):
        self._filename = filename
        self._spectra = list(spectra)
        self.is_recreated = False
        self.closed = False
        self.reads = 0
        charges = {}
        for s in self._spectra:
            c = charges.setdefault(s.precursor_charge, {'id': [], 'precursor_mz': []})
            c['id'].append(s.identifier)
            c['precursor_mz'].append(s.precursor_mz)
        self.spec_info = {'charge': {z: {'id': np.asarray(c['id']),
                                         'precursor_mz': np.asarray(c['precursor_mz'], np.float32)}
                                     for z, c in charges.items()}}

    def read_all_spectra(self):
        self.reads += 1
        for s in sorted(self._spectra, key=lambda s: str(s.identifier)):
            yield s

    def get_version(self):
        return 'null'

    def close(self):
        self.closed = True

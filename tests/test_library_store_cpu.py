"""Host logic of the reader adapter (ann_solo_amd/library_store.py) on
CPU: the batched ``process_spectrum`` device call is answered by the oracle, everything else --
row order, annotation alignment, validity, the on-disk store, the query-side packing of
spectral_library.py:207-228 -- is the product's own code."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_reader import FakeReader, FakeSpectrum, raw_spectrum   # noqa: E402


@pytest.fixture()
def oracle_process(O, monkeypatch):
    """ann_solo_amd.spectrum.process_spectra answered per spectrum by orc_process_spectrum."""
    from ann_solo_amd import spectrum
    from ann_solo_amd.packed import PackedSpectra

    def process_spectra(raw, is_library, config=None, device='cpu'):
        g = lambda k, d: getattr(config, k, d) if config is not None else d
        o, mz, it, chg, pmz, pz = raw.numpy()
        offs, mzs, its, chgs, valid = [0], [], [], [], []
        for s in range(raw.n):
            sl = slice(o[s], o[s + 1])
            ok, rm, ri, src = O.process_spectrum(
                mz[sl], it[sl], pmz[s], pz[s], g('min_mz', 11), g('max_mz', 2010),
                g('remove_precursor', False), g('remove_precursor_tolerance', 0.0),
                g('min_intensity', 0.01),
                g('max_peaks_used_library', 50) if is_library else g('max_peaks_used', 50),
                g('scaling', 'rank'), g('min_peaks', 10), g('min_mz_range', 250.0),
                g('resolution', None))
            valid.append(ok)
            if ok:
                mzs.append(rm), its.append(ri), chgs.append(chg[sl][src])
            offs.append(offs[-1] + (len(rm) if ok else 0))
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
        return (PackedSpectra.from_numpy(np.asarray(offs), cat(mzs, np.float32), cat(its, np.float32),
                                         cat(chgs, np.uint8), pmz, pz, 'cpu', raw.identifiers),
                torch.as_tensor(np.asarray(valid, bool)))
    monkeypatch.setattr(spectrum, 'process_spectra', process_spectra)
    return process_spectra


def _library(rng, n=60, str_ids=False):
    specs = []
    for i in range(n):
        z = int(rng.choice([2, 3], p=[0.7, 0.3]))
        specs.append(raw_spectrum(rng, f'id{i}' if str_ids else 3 * i + 1, z, good=(i % 11 != 5)))
    return specs


def test_store_rows_follow_spec_info_and_peaks_are_processed(O, oracle_process):
    from ann_solo_amd import library_store as ls
    from ann_solo_amd.spectral_library import Config
    rng = np.random.default_rng(3)
    specs = _library(rng)
    reader = FakeReader(specs)
    st = ls.build_library_store(reader, Config(), 'cpu', chunk=17)      # several ragged chunks
    assert reader.reads == 1
    by_id = {s.identifier: s for s in specs}
    o, mz, it, chg, pmz, pz = st.spectra.numpy()
    n_invalid = 0
    for z, (a, b) in st.ranges.items():
        ids = reader.spec_info['charge'][z]['id'].tolist()
        assert [m['identifier'] for m in st.meta[z]] == ids          # row r <-> spec_info id[r]
        for r, ident in enumerate(ids):
            s = by_id[ident]
            row = a + r
            ok, rm, ri, src = O.process_spectrum(s.mz, s.intensity, s.precursor_mz, z)
            assert st.valid[row] == ok and pz[row] == z and pmz[row] == s.precursor_mz
            assert st.meta[z][r]['peptide'] == s.peptide and st.meta[z][r]['is_decoy'] == s.is_decoy
            if not ok:
                n_invalid += 1
                assert o[row + 1] == o[row]
                continue
            sl = slice(o[row], o[row + 1])
            assert np.array_equal(mz[sl], rm) and np.array_equal(it[sl], ri)
            want = np.array([0 if s.annotation[j] is None else s.annotation[j].charge for j in src])
            assert np.array_equal(chg[sl], want)                      # annotations follow their peaks
    assert 0 < n_invalid < len(specs)
    # the snapshot's alignment (raw annotation array restored, reader.py:243-245): peak j <- raw j
    st2 = ls.build_library_store(FakeReader(specs), Config(), 'cpu', 'snapshot')
    o2, _, _, chg2, _, _ = st2.spectra.numpy()
    assert np.array_equal(o2, o) and not np.array_equal(chg2, chg)
    z0, (a0, _) = next(iter(st2.ranges.items()))
    s = by_id[reader.spec_info['charge'][z0]['id'].tolist()[0]]
    k = o2[a0 + 1] - o2[a0]
    assert np.array_equal(chg2[o2[a0]:o2[a0 + 1]],
                          [0 if a is None else a.charge for a in s.annotation[:k]])


def test_store_round_trip_and_invalidation(tmp_path, oracle_process):
    from ann_solo_amd import library_store as ls
    from ann_solo_amd.spectral_library import Config
    rng = np.random.default_rng(4)
    specs = _library(rng, 40, str_ids=True)
    cfg = Config()
    key = ls.store_hash(cfg, 'abc', 'peaks')
    path = str(tmp_path / f'lib_{key[:7]}{ls.STORE_EXT}')
    r1 = FakeReader(specs)
    st = ls.load_or_build_library_store(r1, cfg, 'cpu', path, key)
    assert os.path.isfile(path) and r1.reads == 1
    r2 = FakeReader(specs)
    st2 = ls.load_or_build_library_store(r2, cfg, 'cpu', path, key)
    assert r2.reads == 0                                              # served from the store
    for x, y in zip(st.spectra.numpy(), st2.spectra.numpy()):
        assert np.array_equal(x, y)
    assert np.array_equal(st.valid, st2.valid) and st.ranges == st2.ranges and st.meta == st2.meta
    # other preprocessing options -> other key; a recreated reader or another library -> rebuild
    assert ls.store_hash(Config(min_intensity=0.05), 'abc', 'peaks') != key
    assert ls.store_hash(cfg, 'abd', 'peaks') != key and ls.store_hash(cfg, 'abc', 'snapshot') != key
    r3 = FakeReader(specs)
    r3.is_recreated = True
    ls.load_or_build_library_store(r3, cfg, 'cpu', path, key)
    assert r3.reads == 1
    r4 = FakeReader(specs[:-1])
    st4 = ls.load_or_build_library_store(r4, cfg, 'cpu', path, key)
    assert r4.reads == 1 and st4.spectra.n == len(specs) - 1
    with pytest.raises(ValueError):
        ls.load_library_store(path, 'some-other-key')


def test_pack_queries_mirrors_the_reference_loop(O, oracle_process):
    """spectral_library.py:207-228: unknown charge -> copies at 2 and 3; invalid copies dropped;
    per charge in file order; charges in order of first appearance."""
    from ann_solo_amd import library_store as ls
    from ann_solo_amd.spectral_library import Config
    rng = np.random.default_rng(5)
    qs = []
    for i in range(30):
        z = [3, 2, None, 2, 4][i % 5]
        s = raw_spectrum(rng, f'scan={i}', z, good=(i % 7 != 3))
        s.retention_time = 0.25 * i
        if i % 2:
            s.index = 1000 + i
        qs.append(s)
    packed, meta = ls.pack_queries(iter(qs), Config(), 'cpu', chunk=8)
    assert list(packed) == list(meta) == [3, 2, 4]
    for z in packed:
        want = [(s, i) for i, s in enumerate(qs) if (s.precursor_charge == z or
                                                      (s.precursor_charge is None and z in (2, 3)))
                and O.process_spectrum(s.mz, s.intensity, s.precursor_mz, z)[0]]
        assert [m['identifier'] for m in meta[z]] == [s.identifier for s, _ in want]
        assert [m['index'] for m in meta[z]] == [getattr(s, 'index', i) for s, i in want]
        assert all(m['precursor_charge'] == z for m in meta[z])
        assert packed[z].n == len(want) and (packed[z].precursor_charge == z).all()
        o, mz, it, *_ = packed[z].numpy()
        for r, (s, _) in enumerate(want):
            _, rm, ri, _ = O.process_spectrum(s.mz, s.intensity, s.precursor_mz, z)
            assert np.array_equal(mz[o[r]:o[r + 1]], rm) and np.array_equal(it[o[r]:o[r + 1]], ri)
    unknown = [s.identifier for s in qs if s.precursor_charge is None]
    assert set(unknown) & {m['identifier'] for m in meta[2]} & {m['identifier'] for m in meta[3]}


def test_unsorted_peaks_and_missing_annotations_are_tolerated(oracle_process):
    from ann_solo_amd import library_store as ls
    s = FakeSpectrum('a', 500.0, 2, [300.0, 100.0, 200.0], [3.0, 1.0, 2.0])
    p = ls.pack_raw([s])
    assert p.mz.tolist() == [100.0, 200.0, 300.0] and p.intensity.tolist() == [1.0, 2.0, 3.0]
    assert p.charge.tolist() == [0, 0, 0]
    q = ls.pack_raw([s], charges=[3])
    assert q.precursor_charge.tolist() == [3]
    c = ls.concat_packs([p, q])
    assert c.n == 2 and c.offsets.tolist() == [0, 3, 6] and c.identifiers == ['a', 'a']

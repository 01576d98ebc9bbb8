"""the engine is constructed from the REFERENCE's configuration object and
driven exactly as ``ann_solo.py:76-83`` drives the reference's -- the python block of
INTEGRATION.md 4a is executed verbatim. ``tests/ref_config.RefConfig`` answers like the
reference's ``config`` singleton: ``KeyError`` for any option its parser does not define
(config.py:285-291), e.g. the additive ``index`` / ``pq_m`` / ``num_gpus``.

CPU: the device calls of the engine are answered by the oracle (window search + best match),
everything else is the product's host code. GPU (``-m gpu``): the real engine, IVF-Flat (the
default the reference's own flags select) and IVF-PQ through the additive flags."""
import os
import re
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fake_reader import FakeReader, FakeSpectrum, Annotation   # noqa: E402
from ref_config import RefConfig                                  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ('--precursor_tolerance_mass 20 --precursor_tolerance_mode ppm '
            '--fragment_mz_tolerance 0.02 --allow_peak_shifts')


def _snippet() -> str:
    """The first python block under '## 4a.' of INTEGRATION.md."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = text[text.index('## 4a.'):]
    return re.search(r'```python\n(.*?)```', sec, re.S).group(1)


def _objects(pack, ids=None, blow_up=True):
    o, mz, it, chg, pmz, pz = pack.to('cpu').numpy()
    out = []
    for i in range(pack.n):
        sl = slice(o[i], o[i + 1])
        ann = [Annotation(int(c)) if c else None for c in chg[sl]]
        out.append(FakeSpectrum(ids[i] if ids is not None else i, float(pmz[i]), int(pz[i]), mz[sl],
                                np.exp(6.0 * it[sl]) if blow_up else it[sl], ann, f'PEP{i}K',
                                is_decoy=(i % 9 == 0), retention_time=0.5 * i, index=i))
    return out


def test_reference_config_object_behaviour_and_snapshot():
    from ann_solo_amd.config import Config
    ref = RefConfig()
    with pytest.raises(RuntimeError):           # config.py:286-287
        ref.num_list
    ref.parse(f'/data/lib.splib /data/q.mgf /data/out {REQUIRED} --num_list 64 --scaling sqrt '
              '--precursor_tolerance_mass_open 300 --precursor_tolerance_mode_open Da')
    for name in ('index', 'pq_m', 'pq_bits', 'refine_k', 'kmeans_niter', 'ann_seed', 'num_gpus'):
        with pytest.raises(KeyError):           # config.py:288: self._namespace[option]
            getattr(ref, name)
    cfg = Config.from_reference(ref)
    assert (cfg.index, cfg.pq_m, cfg.pq_bits, cfg.refine_k, cfg.kmeans_niter, cfg.seed,
            cfg.num_gpus) == ('ivfflat', 32, 8, None, 25, 1234, 0)
    assert cfg.num_list == 64 and cfg.scaling == 'sqrt' and cfg.allow_peak_shifts is True
    assert cfg.spectral_library_filename == '/data/lib.splib' and cfg.out_filename == '/data/out'
    assert cfg.precursor_tolerance_mass_open == 300 and cfg.model == 'rf'
    for k in ('min_mz', 'max_mz', 'bin_size', 'hash_len', 'num_candidates', 'batch_size',
              'num_probe', 'fdr', 'fdr_min_group_size', 'mode', 'min_peaks', 'min_mz_range'):
        assert cfg[k] == ref[k], k
    assert Config.from_reference(cfg) is cfg and Config.from_reference(None) == Config()
    assert Config.from_reference({'num_list': 8, 'seed': 3}).seed == 3
    # the additive patch: same parser + add_arguments
    ref2 = RefConfig(additive=True)
    ref2.parse(f'l.splib q.mgf out {REQUIRED} --index ivfpq --pq_m 16 --pq_bits 8 --refine_k 2048 '
               '--kmeans_niter 5 --ann_seed 7 --num_gpus 2')
    cfg2 = Config.from_reference(ref2)
    assert (cfg2.index, cfg2.pq_m, cfg2.refine_k, cfg2.kmeans_niter, cfg2.seed, cfg2.num_gpus) == \
        ('ivfpq', 16, 2048, 5, 7, 2)
    ref3 = RefConfig(additive=True)
    ref3.parse(f'l.splib q.mgf out {REQUIRED}')
    assert Config.from_reference(ref3) == Config.from_reference(
        ref3, index='ivfflat')                  # defaults of the flags == defaults of the dataclass
    assert Config.from_reference(ref3).index == 'ivfflat'
    with pytest.raises(SystemExit):
        RefConfig(additive=True).parse(f'l q o {REQUIRED} --index hnsw')


def _hash_of(cfg):
    from ann_solo_amd.spectral_library import SpectralLibrary
    sl = SpectralLibrary.__new__(SpectralLibrary)
    sl.config = cfg
    return sl._get_hyperparameter_hash(), sl._get_index_hash()


def test_index_file_hash_is_the_reference_hash_for_the_reference_flags():
    """With only the reference's flags the cached index keeps the reference's 7-digit hash."""
    from ann_solo_amd.config import Config
    ref = RefConfig()
    ref.parse(f'l.splib q.mgf out {REQUIRED}')
    h, hi = _hash_of(Config.from_reference(ref))
    assert h == hi
    ref2 = RefConfig(additive=True)
    ref2.parse(f'l.splib q.mgf out {REQUIRED} --index ivfpq')
    h2, hi2 = _hash_of(Config.from_reference(ref2))
    assert h2 == h and hi2 != hi


class _OracleEngineMixin:
    """Device calls of SpectralLibrary answered by the oracle (CPU test only)."""

    def _add_partition(self, z, spectra, ids, valid, pmz32):
        from ann_solo_amd.spectral_library import ChargePartition
        self.partitions[z] = ChargePartition(z, np.asarray(ids), np.ascontiguousarray(pmz32, np.float32),
                                             spectra, None)
        self.partitions[z].valid = np.asarray(valid, bool)

    def _search_batch(self, queries, charge, mode, want_knn=False, device_out=False):
        from oracle import oracle_py as O
        if charge not in self.partitions:
            return None
        part = self.partitions[charge]
        tol, tmode = self._tolerance(mode)
        Q, L = O.Spectra(*queries.numpy()), O.Spectra(*part.spectra.numpy())
        n = Q.n
        stride = max(1, int(np.diff(Q.offsets).max()))
        res = SimpleNamespace(best_row=np.full(n, -1, np.int32), best_score=np.zeros(n),
                              n_candidates=np.zeros(n, np.int32), pm_count=np.zeros(n, np.int32),
                              pm_pairs=np.zeros((n, stride, 2), np.uint32))
        res.peak_matches = lambda i: res.pm_pairs[i, :res.pm_count[i]].astype(np.int64)
        lp = part.precursor_mz.astype(np.float64)
        for i in range(n):
            qp = Q.precursor_mz[i]
            ok = (np.abs(qp - lp) * charge <= tol) if tmode == 'Da' else (np.abs(qp - lp) / lp * 1e6 <= tol)
            cand = np.nonzero(ok & part.valid)[0].astype(np.int64)
            res.n_candidates[i] = len(cand)
            if len(cand) == 0:
                continue
            b, s, m = O.best_match(Q, i, L, cand, self.config.fragment_mz_tolerance,
                                   self.config.allow_peak_shifts)
            if b >= 0:
                res.best_row[i], res.best_score[i], res.pm_count[i] = cand[b], s, len(m)
                res.pm_pairs[i, :len(m)] = m
        return res


def _oracle_cosine(q, lib, rows, pairs, cnt):
    qo, _, qi, *_ = q.numpy()
    lo, _, li, *_ = lib.numpy()
    out = np.zeros(q.n)
    for i in range(q.n):
        if rows[i] >= 0:
            p = pairs[i, :cnt[i]].astype(np.int64)
            out[i] = float(np.sum(qi[qo[i] + p[:, 0]].astype(np.float64) * li[lo[rows[i]] + p[:, 1]]))
    return out


def _run_snippet(config, engine_cls, reader_factory, query_reader, tmp_path, scorer=None,
                 device='cuda'):
    """Execute INTEGRATION.md 4a's block. The names it uses beyond ``config``: ``writer`` (the
    reference's module: here a recorder behind the reference's call signature -- it keeps what
    the block hands to ``write_mztab`` and answers the columns through tests/mztab_check.py) and
    ``score_ssms``; the engine class gets the test's reader seams pre-bound."""
    import os
    import mztab_check as M
    from ann_solo_amd import spectral_library as real
    made = {}

    def make(filename, config=None, score_ssms=None):
        made['sl'] = engine_cls(filename, config=config, score_ssms=score_ssms,
                                reader_factory=reader_factory, query_reader=query_reader,
                                device=device)
        return made['sl']
    module = SimpleNamespace(SpectralLibrary=make)
    def write_mztab(ids, fn, reader):          # writer.py:40-61: signature, '.mztab' suffix rule
        assert isinstance(reader.get_version(), str)
        made['file'] = {'name': fn if os.path.splitext(fn)[1].lower() == '.mztab' else fn + '.mztab',
                        'rows': [M.record_fields(s_) for s_ in sorted(ids, key=lambda s_: M.natural_key(s_.query_identifier))],
                        'settings': {k_: str(config[k_]) for k_ in ('mode', 'num_list', 'num_probe', 'fdr')}}
    writer = SimpleNamespace(write_mztab=write_mztab)
    code = _snippet()
    assert 'from ann_solo_amd import spectral_library' in code
    # the import line is executed for real (the module must import); the name is then re-bound to
    # the same module with the test's reader seams
    env = dict(config=config, writer=writer, score_ssms=scorer)
    exec(code.replace('from ann_solo_amd import spectral_library',
                      'from ann_solo_amd import spectral_library as _real; spectral_library = _shim'),
         dict(env, _shim=module))
    assert real.SpectralLibrary is not None
    return made['sl'], made['file']


def test_engine_from_reference_config_cpu(O, tmp_path, monkeypatch):
    """CPU: reference-shaped config -> SpectralLibrary(filename, config=config) -> search(query
    file) -> mzTab, the snippet of INTEGRATION.md 4a verbatim; brute-force mode (no ANN index)."""
    from ann_solo_amd import spectrum, spectrum_similarity, synthetic
    from ann_solo_amd.packed import PackedSpectra
    from ann_solo_amd.spectral_library import SpectralLibrary

    def process_spectra(raw, is_library, config=None, device='cpu'):
        g = lambda k: getattr(config, k)
        o, mz, it, chg, pmz, pz = raw.numpy()
        offs, mzs, its, chgs, valid = [0], [], [], [], []
        for s in range(raw.n):
            sl = slice(o[s], o[s + 1])
            ok, rm, ri, src = O.process_spectrum(
                mz[sl], it[sl], pmz[s], pz[s], g('min_mz'), g('max_mz'), g('remove_precursor'),
                g('remove_precursor_tolerance'), g('min_intensity'),
                g('max_peaks_used_library') if is_library else g('max_peaks_used'), g('scaling'),
                g('min_peaks'), g('min_mz_range'), g('resolution'))
            valid.append(ok)
            if ok:
                mzs.append(rm), its.append(ri), chgs.append(chg[sl][src])
            offs.append(offs[-1] + (len(rm) if ok else 0))
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
        return (PackedSpectra.from_numpy(np.asarray(offs), cat(mzs, np.float32), cat(its, np.float32),
                                         cat(chgs, np.uint8), pmz, pz, 'cpu', raw.identifiers),
                torch.as_tensor(np.asarray(valid, bool)))
    monkeypatch.setattr(spectrum, 'process_spectra', process_spectra)
    from ann_solo_amd import library_store
    monkeypatch.setattr(library_store, 'process_spectra', process_spectra, raising=False)
    monkeypatch.setattr(spectrum_similarity, 'ssm_cosine', _oracle_cosine)

    class Engine(_OracleEngineMixin, SpectralLibrary):
        pass
    lib, aux = synthetic.make_library(300, seed=11, device='cpu', charges=(2, 3), charge_p=(0.7, 0.3))
    q, truth = synthetic.make_queries(lib, aux, 40, seed=12, open_range=300.0)
    lib_objs = _objects(lib)
    q_objs = _objects(q, ids=[f'scan={i}' for i in range(q.n)])
    fn = str(tmp_path / 'lib.splib')
    config = RefConfig()                         # the reference's flags only
    config.parse(f'{fn} {tmp_path / "q.mgf"} {tmp_path / "out"} {REQUIRED} --mode bf '
                 '--precursor_tolerance_mass_open 300 --precursor_tolerance_mode_open Da '
                 '--batch_size 16')
    sl, out = _run_snippet(config, Engine, lambda f, h: FakeReader(lib_objs, f),
                           lambda f: iter(q_objs), tmp_path, device='cpu')
    assert isinstance(sl, SpectralLibrary) and sl.config.index == 'ivfflat' and sl.config.mode == 'bf'
    assert sl._library_reader.closed and out['name'] == str(tmp_path / 'out') + '.mztab'
    rows = out['rows']
    src = truth['source_row'].numpy()
    assert len(rows) > 0.8 * q.n
    right = sum(int(r['opt_ms_run[1]_cv_MS:1003062_spectrum_index']) == src[int(r['PSM_ID'].split('=')[1])]
                for r in rows)
    assert right > 0.8 * len(rows)
    assert out['settings']['mode'] == 'bf'


def test_num_gpus_flag_needs_a_matching_job(O, monkeypatch):
    """``--num_gpus N`` shards at construction over the N ranks of the torch.distributed job the
    process belongs to; a single process asking for 2 GPUs is told how to launch."""
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import SpectralLibrary

    class Engine(_OracleEngineMixin, SpectralLibrary):
        pass
    lib, _ = synthetic.make_library(50, seed=3, device='cpu', charges=(2,), charge_p=(1.0,))
    config = RefConfig(additive=True)
    config.parse(f'l.splib q.mgf out {REQUIRED} --mode bf --num_gpus 2')
    with pytest.raises(RuntimeError, match='torchrun --nproc-per-node 2'):
        Engine(lib, config=config, device='cpu')
    config.parse(f'l.splib q.mgf out {REQUIRED} --mode bf --num_gpus 1')
    Engine(lib, config=config, device='cpu')          # 0 / 1: no sharding asked for
    # the reference's --no_gpu cannot be honoured (no CPU fallback): refused, not ignored
    from ann_solo_amd._lib import AnnSoloMiError
    config.parse(f'l.splib q.mgf out {REQUIRED} --mode bf --no_gpu')
    with pytest.raises(AnnSoloMiError, match='no CPU fallback'):
        Engine(lib, config=config, device='cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('flags', ['', '--index ivfpq --pq_m 32 --kmeans_niter 4'])
def test_engine_from_reference_config_gpu(tmp_path, flags):
    """GPU: the same block against the real engine. Without additive flags the reference's own
    options select IVF-Flat (FAISS' defaults: 25 iterations, seed 1234), and the cached index is
    named with the reference's hash; with the patch of 4a the additive flags select IVF-PQ."""
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import SpectralLibrary, INDEX_EXT
    lib, aux = synthetic.make_library(5000, seed=41, device='cpu', charges=(2, 3), charge_p=(0.7, 0.3))
    q, truth = synthetic.make_queries(lib, aux, 300, seed=42, open_range=300.0)
    lib_objs = _objects(lib)
    q_objs = _objects(q, ids=[f'scan={i}' for i in range(q.n)])
    fn = str(tmp_path / 'human.splib')
    config = RefConfig(additive=bool(flags))
    config.parse(f'{fn} {tmp_path / "q.mgf"} {tmp_path / "out.mztab"} {REQUIRED} --num_list 32 '
                 f'--num_probe 16 --num_candidates 512 --batch_size 128 '
                 f'--precursor_tolerance_mass_open 300 --precursor_tolerance_mode_open Da {flags}')
    if not flags:
        with pytest.raises(KeyError):
            config.index

    def scorer(ssms, mode):
        for s in ssms:
            s.q = 0.001 if s.search_engine_score > 0.6 else 0.5
        return ssms
    sl, out = _run_snippet(config, SpectralLibrary, lambda f, h: FakeReader(lib_objs, f),
                           lambda f: iter(q_objs), tmp_path, scorer)
    assert sl.config.index == ('ivfpq' if flags else 'ivfflat')
    h7 = sl._get_index_hash()[:7]
    if not flags:
        assert h7 == sl._get_hyperparameter_hash()[:7]
    files = sorted(os.listdir(tmp_path))
    assert [f for f in files if f.endswith(INDEX_EXT)] == [f'human_{h7}_{z}{INDEX_EXT}' for z in (2, 3)]
    rows = out['rows']
    src = truth['source_row'].numpy()
    assert len(rows) > 0.6 * q.n
    right = sum(int(r['opt_ms_run[1]_cv_MS:1003062_spectrum_index']) == src[int(r['PSM_ID'].split('=')[1])]
                for r in rows)
    assert right > 0.9 * len(rows)
    assert out['settings']['num_list'] == '32' and out['settings']['num_probe'] == '16'

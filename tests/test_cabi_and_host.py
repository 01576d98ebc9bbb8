"""CPU-side checks: the C-ABI library loads and exports every symbol include/annsolo_mi.h
declares; host-only entry points agree with the oracle / golden vectors; compute entry
points fail loudly without a GPU (no CPU fallback); host-side containers behave."""
import ctypes as C
import hashlib
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'annsolo_mi.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(asl_[a-z0-9_]+)\s*\(', src)))


def test_every_declared_symbol_is_exported():
    from ann_solo_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 40
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert set(_lib.EXPORTS) == set(names)
    assert b'gfx950' in L.asl_version()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/annsolo_mi.h compiles as strict C99 and a C program
    links against libannsolo_mi.so without any C++ or HIP header."""
    import shutil
    import subprocess
    if not shutil.which('gcc'):
        pytest.skip('gcc not available')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from ann_solo_amd import _lib
    src = os.path.join(tmp_path, 't.c')
    with open(src, 'w') as f:
        f.write('#include <stdio.h>\n#include "annsolo_mi.h"\n'
                'int main(void) { int64_t n; double a, b;\n'
                '  if (asl_get_dim(11, 2010, 0.04, &n, &a, &b)) return 2;\n'
                '  printf("%lld %d %s\\n", (long long)n, asl_hash_idx(12345, 800, 42), asl_version());\n'
                '  return 0; }\n')
    exe = os.path.join(tmp_path, 't')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror',
                           '-I', os.path.join(root, 'include'), src, _lib.LIB_PATH,
                           '-Wl,-rpath,' + os.path.dirname(_lib.LIB_PATH), '-o', exe])
    env = dict(os.environ, LD_LIBRARY_PATH='/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=60)
    assert out.returncode == 0, out.stderr
    n, h, ver = out.stdout.split()[:3]
    assert (int(n), int(h)) == (49976, 584)          # get_dim / hash_idx known answers


def test_host_only_entry_points_match_golden(O, golden):
    from ann_solo_amd import spectrum
    g = golden('encoder_golden.npz')
    assert [spectrum.hash_idx(int(b), 800) for b in g['bins']] == g['hashes'].tolist()
    assert [spectrum.hash_idx(int(b), 64) for b in g['bins']] == g['hashes64'].tolist()
    for args, want in zip(g['dim_args'], g['dims']):
        assert spectrum.get_dim(*args) == (int(want[0]), want[1], want[2])
    assert spectrum.get_dim(11, 2010, 0.04) == (49976, 10.96, 2010.0)


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_no_cpu_fallback():
    from ann_solo_amd import _lib, spectrum
    from ann_solo_amd import faiss_compat as faiss
    assert faiss.get_num_gpus() == 0
    with pytest.raises(_lib.AnnSoloMiError, match='no HIP device'):
        spectrum.spectra_to_vectors(np.array([100.0], np.float32), np.array([1.0], np.float32),
                                    np.array([0, 1], np.int32), 11, 2010, 0.04, 800)
    with pytest.raises(_lib.AnnSoloMiError):
        faiss.IndexFlatIP(800)


def test_config_refuses_what_the_kernels_cannot_hold():
    """Capacity limits the reference does not have are a construction-time ``ValueError`` of
    ``Config``, not an error deep inside a search."""
    from ann_solo_amd.config import Config
    for kw in (dict(max_peaks_used=257), dict(max_peaks_used_library=1000),
               dict(num_candidates=16385), dict(num_probe=2049), dict(flat_storage='fp16')):
        with pytest.raises(ValueError):
            Config(**kw)
    with pytest.raises(ValueError):
        Config.from_reference(dict(max_peaks_used=500))
    assert Config(max_peaks_used=256, num_candidates=2048).max_peaks_used == 256
    assert Config(num_candidates=5000).num_candidates == 5000      # beyond 2 048: bounded passes (index.hip)


def test_hyperparameter_hash_formula():
    """spectral_library.py:118-131: sha1 of the JSON of the five index hyper-parameters."""
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    cfg = Config()
    want = hashlib.sha1(json.dumps({'min_mz': 11, 'max_mz': 2010, 'bin_size': 0.04,
                                    'hash_len': 800, 'num_list': 256}).encode('utf-8')).hexdigest()
    sl = SpectralLibrary.__new__(SpectralLibrary)
    sl.config = cfg
    assert sl._get_hyperparameter_hash() == want
    # reference defaults that reach the hot path (config.py:71-216)
    assert (cfg.num_candidates, cfg.batch_size, cfg.num_list, cfg.num_probe) == (1024, 16384, 256, 128)


def test_packed_spectra_select_and_from_spectra():
    from ann_solo_amd import synthetic
    from ann_solo_amd.packed import PackedSpectra
    lib, aux = synthetic.make_library(200, seed=5, device='cpu')
    rows = torch.tensor([7, 0, 199, 7])
    sub = lib.select(rows)
    o, mz, *_ = lib.numpy()
    so, smz, *_ = sub.numpy()
    for j, r in enumerate(rows.tolist()):
        assert np.array_equal(smz[so[j]:so[j + 1]], mz[o[r]:o[r + 1]])
    assert sub.precursor_mz.tolist() == lib.precursor_mz[rows].tolist()

    class Ann:
        def __init__(self, c):
            self.charge = c

    class Spec:
        pass
    s = Spec()
    s.mz, s.intensity = np.array([100., 200.]), np.array([.6, .8])
    s.precursor_mz, s.precursor_charge, s.annotation = 500.25, 2, [None, Ann(2)]
    p = PackedSpectra.from_spectra([s, s])
    assert p.n == 2 and p.charge.tolist() == [0, 2, 0, 2] and p.offsets.tolist() == [0, 2, 4]


def test_packed_store_round_trip(tmp_path):
    """On-disk processed-peak store: bit-identical round trip, keyed by
    the hyper-parameter hash, corrupt/truncated files rejected."""
    from ann_solo_amd import synthetic
    from ann_solo_amd.packed import PackedSpectra
    lib, _ = synthetic.make_library(300, seed=6, device='cpu')
    lib.identifiers = [f'spec_{i}' for i in range(lib.n)]
    path = os.path.join(tmp_path, 'lib_abc1234.spstore')
    lib.save(path, 'abc1234')
    back = PackedSpectra.load(path, 'abc1234')
    for a, b in zip(lib.numpy(), back.numpy()):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert back.identifiers == lib.identifiers
    assert PackedSpectra.load(path).n == lib.n               # no hash check requested
    with pytest.raises(ValueError):
        PackedSpectra.load(path, 'other')                    # the reference's is_recreated case
    raw = open(path, 'rb').read()
    open(path, 'wb').write(raw[:-100])
    with pytest.raises(ValueError):
        PackedSpectra.load(path)
    open(path, 'wb').write(b'NOTASTORE' + raw[9:])
    with pytest.raises(ValueError):
        PackedSpectra.load(path)
    empty = PackedSpectra.from_numpy([0], [], [], None, [], [])
    empty.save(path)
    assert PackedSpectra.load(path).n == 0


def test_synthetic_generator_contract():
    from ann_solo_amd import synthetic
    a, _ = synthetic.make_library(300, seed=9, device='cpu')
    b, aux = synthetic.make_library(300, seed=9, device='cpu')
    assert torch.equal(a.mz, b.mz) and torch.equal(a.intensity, b.intensity)   # seeded
    o, mz, inten, chg, pmz, pz = a.numpy()
    cnt = np.diff(o)
    assert cnt.min() >= 10 and cnt.max() <= 50 and a.n == 300
    for i in range(a.n):
        s = slice(o[i], o[i + 1])
        assert (np.diff(mz[s]) >= 0).all() and mz[s][-1] - mz[s][0] >= 250
        assert mz[s].min() >= 11 and mz[s].max() <= 2010
        assert abs(np.linalg.norm(inten[s]) - 1) < 1e-5
    q, truth = synthetic.make_queries(b, aux, 64, seed=3)
    assert q.n == 64 and set(truth) == {'source_row', 'is_modified', 'delta_mass'}
    assert (q.charge == 0).all()


def test_search_cascade_control_flow(monkeypatch):
    """SpectralLibrary.search / _search_cascade host logic (spectral_library.py:193-326) with the
    device calls stubbed: batching, first-match-wins per identifier, level 2 only for queries
    level 1 left unidentified, the scorer's q-values gate level 1."""
    from types import SimpleNamespace
    from ann_solo_amd import spectrum_similarity
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    from ann_solo_amd import synthetic
    lib, aux = synthetic.make_library(40, seed=9, device='cpu', charges=(2, 3), charge_p=(0.5, 0.5))
    calls = []

    class Stub(SpectralLibrary):
        def _search_batch(self, queries, charge, mode, want_knn=False, device_out=False):
            calls.append((charge, mode, queries.n))
            n = queries.n
            # std finds a match for even precursor charges*rows only; open finds one for everybody
            found = np.array([(mode == 'open') or (int(round(float(p))) % 2 == 0)
                              for p in queries.precursor_mz])
            return SimpleNamespace(best_row=np.where(found, 1, -1).astype(np.int32),
                                   best_score=np.ones(n), n_candidates=np.ones(n, np.int32),
                                   pm_count=np.ones(n, np.int32), pm_pairs=np.zeros((n, 1, 2), np.uint32),
                                   peak_matches=lambda i: np.zeros((1, 2), np.int64))
    sl = Stub.__new__(Stub)
    sl.config = Config.open_search(batch_size=4)
    sl.device = torch.device('cpu')
    sl.partitions = {2: SimpleNamespace(spectra=lib), 3: SimpleNamespace(spectra=lib)}
    monkeypatch.setattr(spectrum_similarity, 'ssm_cosine',
                        lambda q, l, rows, pairs, cnt: np.full(q.n, 0.5))
    from ann_solo_amd.packed import PackedSpectra
    mk = lambda pm, z: PackedSpectra.from_numpy(np.arange(len(pm) + 1) * 2, np.tile([100., 200.], len(pm)),
                                                np.tile([.6, .8], len(pm)), None, pm, np.full(len(pm), z))
    qs = {2: mk(np.arange(400, 410, dtype=np.float64), 2), 3: mk(np.array([500., 501., 402.]), 3)}
    meta = {2: [dict(identifier=f'scan={i}', index=i, precursor_charge=2, precursor_mz=400. + i)
                for i in range(10)],
            3: [dict(identifier='scan=100', index=100, precursor_charge=3, precursor_mz=500.),
                dict(identifier='scan=101', index=101, precursor_charge=3, precursor_mz=501.),
                dict(identifier='scan=2', index=2, precursor_charge=3, precursor_mz=402.)]}   # charge unknown: tried twice
    lmeta = {z: [dict(identifier=1000 * z + r, peptide=f'PEP{z}{r}K', precursor_mz=300. + r)
                 for r in range(lib.n)] for z in (2, 3)}
    ids = sl.search(qs, meta, lmeta)
    # level 1: batches of 4 over 10 + 3 queries; level 2: only the 5 + 1 odd-m/z queries remain
    assert [c for c in calls if c[1] == 'std'] == [(2, 'std', 4), (2, 'std', 4), (2, 'std', 2), (3, 'std', 3)]
    assert [c for c in calls if c[1] == 'open'] == [(2, 'open', 4), (2, 'open', 1), (3, 'open', 1)]
    by_id = {s.query_identifier: s for s in ids}
    assert len(ids) == 12 and set(by_id) == {m['identifier'] for z in meta for m in meta[z]}
    assert by_id['scan=2'].charge == 2                 # first match wins for the duplicated identifier
    assert all(s.search_engine_score == 0.5 and s.q == 0.0 for s in ids)
    # a scorer that rejects everything at level 1 sends every query to level 2
    calls.clear()

    def scorer(ssms, mode):
        for s in ssms:
            s.q = 1.0 if mode == 'std' else 0.001
        return ssms
    ids = sl.search(qs, meta, lmeta, score_ssms=scorer)
    assert sum(c[2] for c in calls if c[1] == 'open') == 13      # both charge copies of scan=2 retried
    assert len(ids) == 12 and all(s.q == 0.001 for s in ids)


def test_ssm_table_columnar_bookkeeping():
    """SSMTable (the cascade's columnar result): batches appended, first-wins per identifier,
    subsets, concatenation and on-demand SSM records with the writer's attributes."""
    from types import SimpleNamespace
    from ann_solo_amd.spectral_library import SSMTable, _query_uids
    qmeta = {2: [dict(identifier=f'scan={i}', index=i, precursor_charge=2, precursor_mz=400. + i)
                 for i in range(6)],
             3: [dict(identifier=f'scan={i}', index=i, precursor_charge=3, precursor_mz=300. + i,
                      retention_time=1.5) for i in (4, 9)]}
    lmeta = {z: [dict(identifier=100 * z + r, peptide=f'PEP{z}{r}K', precursor_mz=500. + r,
                      is_decoy=(r == 2)) for r in range(5)] for z in (2, 3)}
    mk = lambda rows, cnt: SimpleNamespace(
        pm_count=np.asarray(cnt, np.int32),
        pm_pairs=np.arange(len(rows) * 3 * 2, dtype=np.uint32).reshape(len(rows), 3, 2))
    t = SSMTable(qmeta, lmeta)
    t.add_batch(2, np.array([0, 1, 2]), np.array([4, -1, 2], np.int32), np.array([.9, np.nan, .4]),
                mk([0, 1, 2], [2, 0, 3]))
    t.add_batch(2, np.array([3, 4, 5]), np.array([-1, 0, 1], np.int32), np.array([np.nan, .7, .6]),
                mk([3, 4, 5], [0, 1, 1]))
    t.add_batch(3, np.array([0, 1]), np.array([3, 3], np.int32), np.array([.8, .5]), mk([0, 1], [1, 2]))
    assert len(t) == 6 and t.charge.tolist() == [2, 2, 2, 2, 3, 3] and t.qrow.tolist() == [0, 2, 4, 5, 0, 1]
    assert np.isnan(t.q).all()
    uid = _query_uids(qmeta, [2, 3])
    assert _query_uids(qmeta, [2]) is None
    assert uid[3].tolist() == [uid[2][4], 6]                       # scan=4 is shared, scan=9 is new
    d = t.first_per_uid(uid)                                        # charge 3's scan=4 loses to charge 2's
    assert d.identifiers() == ['scan=0', 'scan=2', 'scan=4', 'scan=5', 'scan=9']
    d.q[:] = [0.0, 0.5, 0.0, 0.5, 0.0]
    keep = d.take(d.q < 0.01)
    assert keep.identifiers() == ['scan=0', 'scan=4', 'scan=9'] and keep.lib_row.tolist() == [4, 0, 3]
    both = SSMTable.concat([keep, d.take(np.array([1]))])
    assert len(both) == 4 and both.batch.tolist() == [0, 1, 2, 3 + 0]   # second table's batches shifted
    s = both[1]
    assert (s.query_identifier, s.library_identifier, s.sequence, s.charge) == ('scan=4', 200, 'PEP20K', 2)
    assert s.search_engine_score == 0.7 and s.q == 0.0 and s.peak_matches.shape == (1, 2)
    assert s.peak_matches.tolist() == [[6, 7]]                      # batch 1, row 1 of its pair table
    last = both[-1]
    assert last.query_identifier == 'scan=2' and last.is_decoy and last.peak_matches.shape == (3, 2)
    assert [x.query_identifier for x in both] == both.identifiers()
    assert both[2].retention_time == 1.5 and both[0].retention_time is None
    ids = both.library_identifiers({2: SimpleNamespace(ids=np.arange(200, 205)),
                                    3: SimpleNamespace(ids=np.arange(300, 305))})
    assert ids.tolist() == [204, 200, 303, 202]


def test_measurement_scripts_and_bench_compile():
    """bench.py, __graft_entry__.py and every helper under scripts/ are at least valid Python (they
    only run on a GPU box), and the shell helpers parse."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py')] + \
        sorted(glob.glob(os.path.join(root, 'scripts', '*.py')))
    assert len(files) > 10
    for f in files:
        compile(open(f).read(), f, 'exec')
    for f in sorted(glob.glob(os.path.join(root, 'scripts', '*.sh'))):
        assert subprocess.run(['bash', '-n', f]).returncode == 0, f


@pytest.mark.parametrize('n', [2, 3])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` WITHOUT torch.distributed.run (no WORLD_SIZE in the environment): the
    process starts the N ranks itself as child processes, relays rank 0's one JSON line and leaves
    with the children's exit code. ASL_BENCH_LAUNCH_CHECK=1 makes the ranks stop after rendezvous + one
    all-reduce (gloo, host tensors): a CPU box has no GPU for the rest."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['ASL_BENCH_LAUNCH_CHECK'] = '1'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '3',
                          '--warmup', '1'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout      # ONE line: nothing else reaches stdout
    rec = json.loads(lines[0])
    assert rec == {'launch_check': True, 'n_gpus': n, 'rank_sum': n * (n + 1) // 2, 'steps': 3, 'warmup': 1}


def test_bench_passes_a_failing_rank_on():
    """A rank that fails makes the self-launched job fail (exit code != 0), not hang or print a line."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    # no launch check: every rank reaches `bench.py needs an MI355X` on this CPU box and exits 1
    if torch.cuda.is_available():
        pytest.skip('CPU-box test')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]


def test_pmc_traffic_is_quoted_only_for_the_kernel_sources_it_was_taken_on(tmp_path, monkeypatch):
    """`roofline.traffic` is a committed constant of a rocprofv3 --pmc pass (counters cannot be read from
    inside the process); bench.py quotes it only while the hash of the scan kernel's sources is the one
    the pass recorded."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    have = bench.kernel_source_sha1('pq')
    assert len(have) == 40 and have != bench.kernel_source_sha1('flat')
    good = tmp_path / 'good.json'
    good.write_text(json.dumps({'kernel_source_sha1': have, 'scan_hbm_bytes_per_launch': 123}))
    stale = tmp_path / 'stale.json'
    stale.write_text(json.dumps({'kernel_source_sha1': '0' * 40, 'scan_hbm_bytes_per_launch': 123}))
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    assert bench._load_traffic('good.json', 'pq')[0] is None          # (sources are read under ROOT: none there)
    monkeypatch.setattr(bench, 'ROOT', ROOT)
    assert bench._load_traffic(os.path.relpath(str(good), ROOT), 'pq') == (123, have)
    t, why = bench._load_traffic(os.path.relpath(str(stale), ROOT), 'pq')
    assert t is None and 'other kernel sources' in why
    assert bench._load_traffic('profiles/does_not_exist.json', 'pq')[0] is None
    # the committed passes: either they belong to these sources (a number) or the line says why not
    for rel, which in ((bench.PMC_TRAFFIC_FILE, 'pq'), (bench.PMC_TRAFFIC_FILE_FLAT, 'flat')):
        t, why = bench._load_traffic(rel, which)
        assert (isinstance(t, int) and t > 0) or (t is None and isinstance(why, str))


def test_bench_stdout_protection_and_the_fallback_line(tmp_path):
    """bench.py's insurance, without a GPU: (1) after protect_stdout() whatever a library writes to file
    descriptor 1 lands on stderr and emit_line() alone reaches stdout; (2) an armed fallback line is printed
    by the watchdog (exit code 0) when the run stops, with the stage it stopped in; (3) ... and by the
    exception path, once only."""
    import subprocess
    import sys
    script = tmp_path / 'fb.py'
    script.write_text(f'''
import os, sys, time
sys.path.insert(0, {ROOT!r})
import bench
mode = sys.argv[1]
bench.protect_stdout()
os.write(1, b"[Gloo] Rank 0 is connected to 1 peer ranks.\\n")      # what a library prints to fd 1
print("python-level print")                                          # sys.stdout writes to fd 1 too
if mode == "line":
    bench.emit_line('{{"ok": true}}')
elif mode == "watchdog":
    bench._arm_fallback(0, 1.0, {{"value": 7, "config": {{"parallelism": "replicas x2"}}}})
    bench._FALLBACK["stage"] = "timed steps"
    time.sleep(30)
elif mode == "raise":
    bench._arm_fallback(0, 0, {{"value": 7}})
    assert bench._emit_fallback("RuntimeError: boom") is True
    assert bench._emit_fallback("again") is False
elif mode == "other-rank":
    bench._arm_fallback(1, 0, {{"value": 7}})
    assert bench._emit_fallback("RuntimeError: boom") is True        # armed, but only rank 0 prints
''')
    def run(mode):
        return subprocess.run([sys.executable, str(script), mode], capture_output=True, text=True, timeout=120)
    out = run('line')
    assert out.returncode == 0 and out.stdout == '{"ok": true}\n', (out.stdout, out.stderr[-500:])
    assert '[Gloo]' in out.stderr and 'python-level print' in out.stderr
    out = run('watchdog')
    assert out.returncode == 0, out.stderr[-500:]
    rec = json.loads(out.stdout)
    assert rec['value'] == 7 and 'watchdog' in rec['sharded_path_failed'] and 'timed steps' in rec['sharded_path_failed']
    out = run('raise')
    assert out.returncode == 0 and json.loads(out.stdout)['sharded_path_failed'] == 'RuntimeError: boom'
    out = run('other-rank')
    assert out.returncode == 0 and out.stdout == ''


def test_bench_help_prints():
    """`python bench.py --help` (argparse expands % in help strings: a bare one raised ValueError)."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0 and '--beyond-llc-chunks' in out.stdout, out.stderr[-500:]

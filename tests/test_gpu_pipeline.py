"""Two-stream software pipeline of the hot path (asl_set_pipeline): batches issued
back to back without waiting -- front of batch i+1 on one stream, scan + rescoring of batch i on
another -- must return exactly what the synchronous path returns, whatever the batch sizes, and
any other entry point must first wait for the batches in flight."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def world():
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    dev = torch.device('cuda', 0)
    lib, aux = synthetic.make_library(60000, seed=11, device=dev, charges=(2,), charge_p=(1.0,))
    out = {}
    for index in ('ivfpq', 'ivfflat'):
        cfg = Config.open_search(num_list=256, num_probe=32, num_candidates=512, index=index, kmeans_niter=4,
                     precursor_tolerance_mass_open=500.0, precursor_tolerance_mode_open='Da')
        out[index] = SpectralLibrary(lib, config=cfg, device=dev)
    batches = [synthetic.make_queries(lib, aux, n, seed=20 + i, open_range=500.0, charge=2)[0].contiguous()
               for i, n in enumerate((2048, 777, 4096, 1, 3000))]
    yield out, batches
    for sl in out.values():
        sl.set_pipeline(False)
        sl.shutdown()


def _same(a, b):
    return all(torch.equal(getattr(a, f), getattr(b, f))
               for f in ('best_row', 'best_score', 'n_candidates', 'pm_count', 'pm_pairs'))


@pytest.mark.parametrize('index', ['ivfpq', 'ivfflat'])
def test_pipelined_batches_equal_synchronous(world, index):
    engines, batches = world
    sl = engines[index]
    sl.set_pipeline(False)
    ref = [sl._search_batch(q, 2, 'open', device_out=True) for q in batches]
    ref_knn = sl._search_batch(batches[0], 2, 'open', want_knn=True, device_out=True)
    torch.cuda.synchronize()
    sl.set_pipeline(True)
    order = [0, 1, 2, 3, 4, 2, 0, 4, 1, 3, 0, 0]
    got = [sl._search_batch(batches[i], 2, 'open', device_out=True) for i in order]   # no waiting
    knn = sl._search_batch(batches[0], 2, 'open', want_knn=True, device_out=True)
    sl.synchronize()
    for i, g in zip(order, got):
        assert _same(g, ref[i]), i
    assert _same(knn, ref_knn) and torch.equal(knn.knn, ref_knn.knn)
    # peak-match rows are fully written: zeros beyond the matches
    pm = got[0].pm_pairs
    slot = torch.arange(pm.shape[1], device=pm.device).unsqueeze(0)
    assert (pm[slot >= got[0].pm_count.unsqueeze(1)] == 0).all()


def test_other_entry_points_wait_for_batches_in_flight(world):
    engines, batches = world
    sl = engines['ivfpq']
    sl.set_pipeline(False)
    ref_open = sl._search_batch(batches[2], 2, 'open', device_out=True)
    ref_std = sl._search_batch(batches[2], 2, 'std')
    ref_host = sl._search_batch(batches[1], 2, 'open')
    vec = sl._encode(batches[4])
    idx = sl._get_ann_index(2)
    D0, I0 = idx.search(vec, 100)
    torch.cuda.synchronize()
    sl.set_pipeline(True)
    a = sl._search_batch(batches[2], 2, 'open', device_out=True)
    b = sl._search_batch(batches[2], 2, 'open', device_out=True)
    std = sl._search_batch(batches[2], 2, 'std')                 # window search: synchronous path
    assert np.array_equal(std.best_row, ref_std.best_row) and np.array_equal(std.best_score, ref_std.best_score)
    c = sl._search_batch(batches[2], 2, 'open', device_out=True)
    host = sl._search_batch(batches[1], 2, 'open')                # host outputs: synchronous path
    assert np.array_equal(host.best_row, ref_host.best_row)
    d = sl._search_batch(batches[2], 2, 'open', device_out=True)
    D1, I1 = idx.search(vec, 100)                                 # shares the index scratch buffers
    assert torch.equal(I1, I0) and torch.equal(D1, D0)
    sl.synchronize()
    for r in (a, b, c, d):
        assert _same(r, ref_open)
    sl.set_pipeline(False)
    assert _same(sl._search_batch(batches[2], 2, 'open', device_out=True), ref_open)


def test_callers_may_drop_their_tensors_before_the_batches_finish(world):
    """The pipeline's streams are invisible to PyTorch's allocator: the engine must keep the
    arrays of a pipelined call alive itself. Found by scripts/fuzz_paths.py -- a temporary query
    pack released right after the call was handed out again and overwritten while the kernels
    of that call were still reading its offsets (memory access fault)."""
    engines, batches = world
    sl = engines['ivfpq']
    sl.set_pipeline(False)
    ref = [sl._search_batch(q, 2, 'open', device_out=True) for q in batches[:3]]
    torch.cuda.synchronize()
    want = [(r.best_row.cpu(), r.best_score.cpu()) for r in ref]
    host = [q.to('cpu') for q in batches[:3]]
    sl.set_pipeline(True)
    got = []
    for rep in range(4):
        for i, q in enumerate(host):
            r = sl._search_batch(q.to('cuda'), 2, 'open', device_out=True)   # temporary device pack
            got.append((i, r.best_row, r.best_score))
            del r
            junk = [torch.full((n,), -7, dtype=torch.int32, device='cuda')     # reuse freed blocks
                    for n in (q.n + 1, q.mz.numel(), q.mz.numel(), 4 * q.n)]
            del junk
    sl.synchronize()
    for i, row, score in got:
        assert torch.equal(row.cpu(), want[i][0]) and torch.equal(score.cpu(), want[i][1])
    sl.set_pipeline(False)


def test_profile_levels_scope_the_stage_events(world):
    """asl_profile_enable(1) brackets every stage with HIP events, (2) only the list scan (what
    bench.py keeps inside its timed region), (0) nothing; the batch results do not depend on it."""
    import ctypes as C
    from ann_solo_amd import _lib
    engines, batches = world
    sl = engines['ivfpq']
    sl.set_pipeline(True)
    L = _lib.lib()

    def launches(name):
        ms, n = C.c_double(), C.c_int64()
        L.asl_profile_get(name.encode(), C.byref(ms), C.byref(n))
        return n.value, ms.value
    ref = None
    try:
        for level, want in ((2, {'scan'}), (1, {'scan', 'encode', 'rescore', 'rescore_matches', 'coarse_select'}),
                            (0, set())):
            L.asl_profile_reset()
            L.asl_profile_enable(level)
            r = sl._search_batch(batches[0], 2, 'open', device_out=True)
            sl.synchronize()
            L.asl_profile_enable(0)
            for name in ('scan', 'encode', 'rescore', 'rescore_matches', 'coarse_select'):
                n, ms = launches(name)
                assert (n > 0) == (name in want), (level, name, n)
                assert ms > 0 or n == 0
            assert ref is None or _same(r, ref)
            ref = r
    finally:
        L.asl_profile_enable(0)
        sl.set_pipeline(False)

"""The N>1 path on CPU: two gloo ranks, inverted lists sharded by asl_lpt_owner, one
exchange of per-shard top-k, merge, data-parallel rescoring -- must reproduce the
unsharded result exactly (compute answered by the oracle backend, tests/oracle_backend.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _widen(q, extra):
    """Queries with ``extra[i]`` additional weak peaks each (so that spectra exceed the 50 peaks
    of the reference's default ``max_peaks_used``; exactly ``50 + extra[i]`` for the widest)."""
    from ann_solo_amd.packed import PackedSpectra
    o, mz, it, chg, pmz, pz = q.numpy()
    rng = np.random.default_rng(5)
    offs, mzs, its, chgs = [0], [], [], []
    for i in range(q.n):
        sl = slice(o[i], o[i + 1])
        n_add = 50 + extra[i] - (o[i + 1] - o[i])
        add = np.sort(rng.uniform(150, 1400, n_add)).astype(np.float32)
        m = np.concatenate([mz[sl], add])
        order = np.argsort(m, kind='stable')
        mzs.append(m[order])
        its.append(np.concatenate([it[sl], np.full(n_add, 0.01, np.float32)])[order])
        chgs.append(np.concatenate([chg[sl], np.zeros(n_add, np.uint8)])[order])
        offs.append(offs[-1] + len(m))
    return PackedSpectra.from_numpy(np.asarray(offs), np.concatenate(mzs), np.concatenate(its),
                                    np.concatenate(chgs), pmz, pz)


def _worker(rank, world, port, kind, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import oracle_py as O
    from oracle_backend import OracleShardBackend
    from ann_solo_amd import synthetic
    from ann_solo_amd import distributed
    from ann_solo_amd.distributed import lpt_owner, sharded_search_batch
    if kind.endswith('_direct'):
        # the RCCL forms of the collectives (all_to_all_single / all_gather_into_tensor on the tensors
        # as they are, asynchronous handles, four pieces per batch) over gloo: no multi-GPU node runs
        # them before the driver's scaling bench does
        distributed.FORCE_DIRECT_COLLECTIVES = True
        kind = kind[:-len('_direct')]
    lib, aux = synthetic.make_library(1500, seed=91, device='cpu', charges=(2,), charge_p=(1.0,))
    q_all, _ = synthetic.make_queries(lib, aux, 2 * 24, seed=92, charge=2)
    lib_np = lib.numpy()
    xb = O.encode_batch(lib_np[1], lib_np[2], lib_np[0], 10.96, 0.04, 800)
    cen = O.kmeans(xb, 8, 4, 1234, 0, 256)
    a = O.assign(xb, cen, 0)
    if kind.startswith('pq'):
        cb = O.pq_train(xb, cen, 8, 16, 4, 1241)
        payload = O.pq_encode(xb, cen, a, cb)
    else:      # 'flat', 'flat_wide'
        cb, payload = None, xb
    pmz32 = lib_np[4].astype(np.float32)
    be = OracleShardBackend(lib_np, pmz32, cen, a, payload, cb, rank, world, 2, 64, 4, 300, 'Da',
                            0.02, True, lpt_owner)
    nloc = q_all.n // world
    wide = kind == 'flat_wide'
    if kind == 'flat_wide':
        # slices whose widest spectra differ (rank 0: <= 70 peaks -> 80-wide rows,
        # rank 1: <= 90 -> 96) must still agree on ONE row width for the peak all-gather
        q_all = _widen(q_all, [20] * nloc + [40] * (q_all.n - nloc))
    q = q_all.select(torch.arange(rank * nloc, (rank + 1) * nloc))
    if kind == 'flat_wide':
        from ann_solo_amd.distributed import _all_gather_peaks, _unpack_peaks
        assert q.max_peaks() == (70 if rank == 0 else 90)
        rows, w = _all_gather_peaks(q, world)
        if w is not None:
            w.wait()
        assert rows.shape == (q_all.n, 2 * 96 + 1)
        back = _unpack_peaks(rows, q)
        assert torch.equal(back.mz, q_all.mz) and torch.equal(back.offsets, q_all.offsets.to(torch.int32))
        rows2, w2 = _all_gather_peaks(q, world, agreed=100)     # a caller-guaranteed common bound
        if w2 is not None:
            w2.wait()
        assert rows2.shape == (q_all.n, 2 * 112 + 1)
        try:
            _all_gather_peaks(q, world, agreed=64)
            raise AssertionError('a spectrum wider than the agreed row must be refused')
        except ValueError:
            pass
    # unsharded reference for my slice
    D, I = be.full.search(be.encode(q).numpy(), 64, 4)
    ref = OracleShardBackend(lib_np, pmz32, cen, a, payload, cb, 0, 1, 2, 64, 4, 300, 'Da', 0.02,
                             True, lpt_owner).rescore_knn(q, torch.from_numpy(I))
    from ann_solo_amd.distributed import CommLog
    # every exchange the driver knows must give the unsharded rows: the two-phase exchange with
    # the default head (at world 2: ceil(2k / 2) = k keys, nothing is ever held back), with small
    # heads (40 of k = 64 keys per shard: the bound / held-back-keys round runs; 1 key per shard:
    # the owner sees fewer than k keys and asks for everything), with phase-2 buffers too small (the
    # flag repeats the batch with the full exchange), the full packed-key rows, the (D, I) rows
    # Round 5, shard-side k: the shards keep rows of shard_keys < k keys; a full row may have dropped
    # keys, and where the bound a shard is sent lies below the row's smallest key it searches that
    # query again with the full k and answers from that row. 'third': rows of 16 keys from 2-4
    # shards cannot even fill k = 64 -- B = 0, every full row is searched again; 'third_mid': some
    # rows are; 'third_overflow': room for ONE second search per piece sends the batch down the
    # full exchange. 'disagree': ONE rank's shard cannot emit packed keys (an empty / dense
    # IVF-Flat shard): all ranks must take the (D, I) rows.
    ok, seen = True, {}
    for name, kw in (('two_phase', {}), ('small_heads', dict(head_keys=40, extras_per_query=32)),
                     ('tiny_heads', dict(head_keys=1, extras_per_query=64)),
                     ('overflow', dict(head_keys=9, extras_per_query=0)),
                     ('third', dict(head_keys=8, shard_keys=16, extras_per_query=64)),
                     ('third_mid', dict(head_keys=20, shard_keys=44, extras_per_query=64)),
                     ('third_overflow', dict(head_keys=8, shard_keys=16, extras_per_query=64)),
                     ('full_keys', dict(two_phase=False)), ('dense_queries', dict(entry_lists=False)),
                     ('disagree', {}), ('rows', None)):
        be.keys = kw is not None and (name != 'disagree' or rank == 0)
        be.rescan_capacity = 1 if name == 'third_overflow' else None
        be.index_epoch = getattr(be, 'index_epoch', 0) + 1     # the agreed exchange format is cached per epoch
        stats, comm = {}, CommLog()
        res = sharded_search_batch(be, q, stats=stats, comm=comm, **(kw or {}))
        same = (np.array_equal(np.sort(res['knn'], 1), np.sort(I, 1)) and
                np.array_equal(res['best_row'], ref['best_row'])
                and np.array_equal(res['best_score'], ref['best_score']))
        ok = ok and same
        seen[name] = (same, stats.get('fallback', 0), sorted(comm.calls), stats.get('third_phase_queries', 0),
                      stats.get('shard_k'), stats.get('exchange_used'), stats.get('query_form'))
    be.keys, be.rescan_capacity = True, None
    ok = (ok and seen['overflow'][1] == 1 and seen['small_heads'][1] == 0 and
          'held_back_keys_all_to_all' in seen['small_heads'][2] and
          ('heads_all_to_all' in seen['two_phase'][2]) == (world > 2) and    # two ranks: the rows travel whole
          seen['third'][3] > nloc and seen['third'][4] == 16 and seen['third'][1] == 0 and
          seen['third'][5] == 'two-phase, shard-side k + second scans' and
          seen['third_mid'][4] == 44 and seen['third_mid'][1] == 0 and seen['third_mid'][3] > 0 and
          seen['small_heads'][3] == 0 and
          seen['third_overflow'][1] == 1 and seen['third_overflow'][5] == 'full rows (fallback)' and
          seen['disagree'][2].count('topk_rows_all_to_all') == 1 and seen['disagree'][5] == 'full rows' and
          'heads_all_to_all' not in seen['disagree'][2] and
          seen['rows'][2].count('topk_rows_all_to_all') == 1 and
          # the other ranks' queries travel as peaks and are hashed straight into entry lists whenever the
          # packed-key scans run (flat_wide: spectra of up to 90 peaks -> dense rows)
          seen['two_phase'][6] == ('dense rows' if wide else 'entry lists') and
          seen['third_mid'][6] == seen['two_phase'][6] and seen['dense_queries'][6] == 'dense rows' and
          seen['disagree'][6] == 'dense rows' and seen['rows'][6] == 'dense rows')
    if not ok:
        print('exchange modes:', seen, flush=True)
    owner_ok = set(be.owner.tolist()) == set(range(world))
    with open(os.path.join(out_dir, f'rank{rank}.txt'), 'w') as f:
        f.write(f'{int(ok)} {int(owner_ok)} {int((res["best_row"] >= 0).sum())}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('kind', ['pq', 'flat', 'flat_wide', 'pq4', 'pq_direct', 'flat_wide_direct', 'pq4_direct'])
def test_two_rank_sharded_search_equals_unsharded(tmp_path, kind):
    """(kind 'pq4': the same through FOUR ranks -- heads really hold keys back by default;
    '*_direct': the collectives in their RCCL form)"""
    world = 4 if kind.startswith('pq4') else 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, kind, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok, owner_ok, n = open(os.path.join(tmp_path, f'rank{r}.txt')).read().split()
        assert ok == '1' and owner_ok == '1' and int(n) > 0


def _replica_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import oracle_py as O
    from oracle_backend import OracleShardBackend
    from ann_solo_amd import synthetic
    from ann_solo_amd.distributed import lpt_owner, make_shard_groups, sharded_search_batch
    # degree == world: the default group, shard rank == rank
    g_full, r_full, gi_full = make_shard_groups(world)
    # degree 1: every rank its own group of one, a full replica serving its own queries
    group, srank, gidx = make_shard_groups(1)
    lib, aux = synthetic.make_library(800, seed=93, device='cpu', charges=(2,), charge_p=(1.0,))
    q_all, _ = synthetic.make_queries(lib, aux, 2 * 12, seed=94, charge=2)
    lib_np = lib.numpy()
    xb = O.encode_batch(lib_np[1], lib_np[2], lib_np[0], 10.96, 0.04, 800)
    cen = O.kmeans(xb, 8, 4, 1234, 0, 256)
    a = O.assign(xb, cen, 0)
    pmz32 = lib_np[4].astype(np.float32)
    be = OracleShardBackend(lib_np, pmz32, cen, a, xb, None, srank, 1, 2, 64, 4, 300, 'Da',
                            0.02, True, lpt_owner)
    nloc = q_all.n // world
    q = q_all.select(torch.arange(rank * nloc, (rank + 1) * nloc))
    res = sharded_search_batch(be, q, group=group)
    D, I = be.full.search(be.encode(q).numpy(), 64, 4)
    ref = be.rescore_knn(q, torch.from_numpy(I))
    ok = (np.array_equal(res['best_row'], ref['best_row'])
          and np.array_equal(res['best_score'], ref['best_score']))
    # the data-plane form bench.py uses at N > 1 (default group = control plane, the shard groups on
    # another backend; here both are gloo): degree == world hands back the caller's all-ranks group,
    # smaller degrees create their groups on the named backend
    g_all = dist.new_group(list(range(world)), backend='gloo')
    g2, r2, gi2 = make_shard_groups(world, backend='gloo', world_group=g_all)
    g3, r3, gi3 = make_shard_groups(1, backend='gloo', world_group=g_all)
    t = torch.tensor([rank + 1])
    dist.all_reduce(t, group=g2)
    planes_ok = (g2 is g_all and r2 == rank and gi2 == 0 and dist.get_world_size(g3) == 1 and r3 == 0
                 and gi3 == rank and int(t) == world * (world + 1) // 2)
    shape_ok = (g_full is None and r_full == rank and gi_full == 0 and srank == 0
                and gidx == rank and dist.get_world_size(group) == 1 and planes_ok)
    with open(os.path.join(out_dir, f'rank{rank}.txt'), 'w') as f:
        f.write(f'{int(ok)} {int(shape_ok)}')
    dist.barrier()
    dist.destroy_process_group()


def test_replica_groups_need_no_exchange(tmp_path):
    """shard degree 1 in a world of 2: two replica groups of one rank each"""
    world = 2
    mp.spawn(_replica_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(tmp_path, f'rank{r}.txt')).read().split() == ['1', '1']


def test_pick_shard_degree():
    from ann_solo_amd.distributed import pick_shard_degree
    gb = 1 << 30
    assert pick_shard_degree(1 * gb, 2 * gb, 8) == 1            # bench library: replicas only
    assert pick_shard_degree(400 * gb, 10 * gb, 8) == 2
    assert pick_shard_degree(1500 * gb, 10 * gb, 8) == 8
    assert pick_shard_degree(5000 * gb, 10 * gb, 8) == 8         # nothing fits: shard fully
    assert pick_shard_degree(100 * gb, 0, 1) == 1


def test_lpt_owner_rule():
    from ann_solo_amd.distributed import lpt_owner
    sizes = np.array([5, 9, 1, 7, 7, 3])
    owner = lpt_owner(sizes, 2)
    # 9->r0, 7(list3)->r1, 7(list4)->r1?? loads: r0=9,r1=7 -> list4 to r1 (14); 5->r0 (14); 3->r0? tie -> r0
    assert owner.tolist() == [0, 0, 1, 1, 1, 0]
    loads = [sizes[owner == r].sum() for r in range(2)]
    assert max(loads) - min(loads) <= sizes.max()
    assert lpt_owner(sizes, 1).tolist() == [0] * 6
    assert lpt_owner(np.zeros(0, np.int64), 4).tolist() == []


# ------------------------------------------------------------------ configs[4]: the cascade
def _cascade_worker(rank, world, port, out_dir, direct=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import oracle_py as O
    from oracle_backend import OracleSpectralLibrary, oracle_cosines
    from ann_solo_amd import distributed, synthetic, spectrum_similarity
    from ann_solo_amd.spectral_library import Config
    distributed.FORCE_DIRECT_COLLECTIVES = bool(direct)     # the collectives in their RCCL form
    spectrum_similarity.ssm_cosine = oracle_cosines
    lib, aux = synthetic.make_library(1400, seed=95, device='cpu', charges=(2, 3),
                                      charge_p=(0.8, 0.2))
    cfg = Config.open_search(batch_size=16, precursor_tolerance_mass_open=300.0,
                 precursor_tolerance_mode_open='Da', fdr=0.01)
    parts, qs, qmeta, lmeta = {}, {}, {}, {}
    pz = lib.precursor_charge.numpy()
    n_q = {2: 37, 3: 5}                                   # ragged against batch 16 and 2 ranks
    for z in (2, 3):
        rows = np.nonzero(pz == z)[0]
        sub = lib.select(torch.as_tensor(rows))
        lib_np = sub.numpy()
        p = dict(lib_np=lib_np, pmz32=lib_np[4].astype(np.float32))
        if z == 2:                                        # charge 3: too rare for an index
            xb = O.encode_batch(lib_np[1], lib_np[2], lib_np[0], 10.96, 0.04, 800)
            cen = O.kmeans(xb, 8, 4, 1234, 0, 256)
            a = O.assign(xb, cen, 0)
            cb = O.pq_train(xb, cen, 8, 16, 4, 1241)
            p.update(centroids=cen, assign=a, codebooks=cb, payload=O.pq_encode(xb, cen, a, cb))
        parts[z] = p
        q, _ = synthetic.make_queries(lib, aux, n_q[z], seed=96 + z, charge=z, open_range=300.0)
        qs[z] = q
        qmeta[z] = [dict(identifier=f'scan={100 * z + i}', index=100 * z + i, precursor_charge=z,
                         precursor_mz=float(q.precursor_mz[i])) for i in range(q.n)]
        lmeta[z] = [dict(identifier=int(1000 * z + r), peptide=f'PEP{z}x{r}K',
                         precursor_mz=float(pm)) for r, pm in enumerate(p['pmz32'])]
    qmeta[3][1]['identifier'] = qmeta[2][4]['identifier']   # unknown charge: one id, tried twice

    def scorer(ssms, mode):                # stands for utils.score_ssms: deterministic q-values
        for s in ssms:
            s.q = 0.001 if s.search_engine_score > 0.6 else 0.5
        return ssms
    one = OracleSpectralLibrary(parts, cfg, 64, 4)
    ref = one.search(qs, qmeta, lmeta, score_ssms=scorer)
    many = OracleSpectralLibrary(parts, cfg, 64, 4, world, rank)
    got = many.search(qs, qmeta, lmeta, score_ssms=scorer)
    key = lambda s: (s.query_identifier, s.library_identifier, s.charge, s.search_engine_score,
                     s.q, s.peak_matches.tolist())
    same = sorted(map(key, ref)) == sorted(map(key, got))
    n_std = sum(s.q < cfg.fdr for s in ref)
    # the second level really ran sharded: shard 0 and shard 1 own different lists
    owners = set(many._shard[2].owner.tolist())
    with open(os.path.join(out_dir, f'rank{rank}.txt'), 'w') as f:
        f.write(f'{int(same)} {len(ref)} {n_std} {len(owners)}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('direct', [False, True])
def test_two_rank_sharded_cascade_equals_unsharded(tmp_path, direct):
    """configs[4] on CPU: standard search (data-parallel window search) -> FDR gate -> open search
    of the remainder over the list-sharded index, ragged batches, a charge without an index and
    a duplicated identifier -- identical identifications on every rank and to one process
    (reference: spectral_library.py:237-259, 301-317)."""
    world = 2
    mp.spawn(_cascade_worker, args=(world, _free_port(), str(tmp_path), direct), nprocs=world, join=True)
    seen = []
    for r in range(world):
        same, n, n_std, owners = open(os.path.join(tmp_path, f'rank{r}.txt')).read().split()
        assert same == '1' and owners == '2'
        assert int(n) >= 35 and 5 < int(n_std) < int(n)       # both cascade levels contributed
        seen.append((n, n_std))
    assert seen[0] == seen[1]

"""Functional check of the sharded (multi-GPU) bench path on a single GPU: two ranks, both on
GPU 0, host-side (gloo) collectives -- list sharding, probe-list all-gather,
search_preassigned, top-k exchange + merge and rescoring must reproduce the unsharded
result exactly. RCCL itself is exercised only by the driver's multi-GPU runs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('degree,index,extra', [
    (0, 'ivfpq', []), (1, 'ivfpq', []), (0, 'ivfflat', []),
    (0, 'ivfpq', ['--head-keys', '700']),                             # phase 2 of the exchange runs
    (0, 'ivfflat', ['--head-keys', '600', '--extras-per-query', '0']),   # ... and overflows: fallback
    (0, 'ivfpq', ['--head-keys', '200', '--shard-keys', '448', '--extras-per-query', '1024']),   # 2 x 448 < k: B = 0,
                                                                      # every full row is scanned a second time
    (0, 'ivfflat', ['--head-keys', '600', '--shard-keys', '768', '--extras-per-query', '512']),   # a few rows are
    (0, 'ivfpq', ['--exchange', 'full'])])
def test_two_rank_sharded_bench_path(degree, index, extra):
    """degree 0 = lists sharded over both ranks; degree 1 = two replicas (no exchange). The
    exchange: two-phase by default (at two ranks a head holds the whole row), with smaller heads
    (the owners ask for held-back keys), with no room for the answers (one fallback to the full
    exchange per step), and the full rows."""
    env = dict(os.environ, ASL_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port',
           str(29517 + degree + (5 if index == 'ivfflat' else 0)),
           os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '1', '--warmup', '1', '--library-size', '60000', '--nlist',
           '256', '--niter', '4', '--batch', '1024', '--recall-queries', '64',
           '--shard-degree', str(degree), '--index', index] + extra
    cmd[cmd.index('--master-port') + 1] = str(int(cmd[cmd.index('--master-port') + 1]) + 20 * len(extra))
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert line, out.stderr[-2000:]
    d = json.loads(line[-1])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 2048
    assert d['shard_check']['sharded_equals_unsharded'] is True
    assert d['value'] > 0 and d['roofline']['achieved'] > 0
    if degree == 0:
        assert d['config']['parallelism'] == 'ivf-list-shard x2'
        assert d['alt_layouts']['replicas_only']['value'] > 0
        c = d['comm']
        assert c['total_bytes_out_per_rank_per_step'] > 0 and c['collective_ms_alone']
        if '--exchange' in extra:
            assert 'topk_rows_all_to_all' in c and 'heads_all_to_all' not in c
        elif '--shard-keys' in extra:
            assert c['shard_k'] == int(extra[3]) and c['head_width'] == int(extra[1]) + 1
            if int(extra[3]) * 2 < 1024:
                # two rows cannot fill k: every full row would be scanned again -- more than a piece has
                # room for (rows / 16): the batch falls back to k-deep rows, results still identical
                assert c['fallbacks_to_full_exchange'] == 1 and c['exchange'] == 'full rows (fallback)'
            else:
                assert c['exchange'] == 'two-phase, shard-side k + second scans'
                assert c['fallbacks_to_full_exchange'] == 0 and c['third_phase_queries'] >= 0
        elif '--head-keys' in extra:
            assert c['exchange'].startswith('two-phase' if '--extras-per-query' not in extra else 'full rows')
            assert 'heads_all_to_all' in c and 'held_back_keys_all_to_all' in c
            assert c['fallbacks_to_full_exchange'] == (1 if '--extras-per-query' in extra else 0)
            assert c['head_width'] == int(extra[1]) + 1
        else:       # two ranks: a head of ceil(2k / 2) keys is the whole row -- the rows travel as they are
            assert 'topk_rows_all_to_all' in c and 'heads_all_to_all' not in c
    else:
        assert d['config']['parallelism'] == 'replicas x2' and d['alt_layouts'] is None


@pytest.mark.parametrize('how', ['raise', 'hang'])
def test_a_failing_sharded_path_still_yields_the_replicas_line(how):
    """Insurance for a first multi-GPU lease: once the replicas layout is measured, a sharded path that
    raises -- or sits in a collective until the watchdog fires -- ends the run with THAT measurement as
    the one JSON line (`sharded_path_failed` says why) and exit code 0."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(ASL_BENCH_BACKEND='gloo', ASL_BENCH_INJECT=how)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '1', '--warmup', '1', '--library-size', '60000', '--nlist',
           '256', '--niter', '4', '--batch', '1024', '--recall-queries', '64', '--sharded-watchdog-seconds', '15']
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(line) == 1, out.stderr[-2000:]
    d = json.loads(line[0])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['config']['parallelism'] == 'replicas x2'
    assert ('injected' if how == 'raise' else 'watchdog') in d['sharded_path_failed']


def test_plain_bench_command_starts_two_ranks():
    """`python bench.py --gpus 2 ...` with no launcher in front: bench.py starts the two ranks itself
    (children of torch.distributed.run; the parent never touches the GPU) and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['ASL_BENCH_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '1', '--warmup', '1', '--library-size', '60000', '--nlist',
           '256', '--niter', '4', '--batch', '1024', '--recall-queries', '64']
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(line) == 1 and line[0].startswith('{'), (out.stdout[:500], out.stderr[-2000:])   # nothing else on stdout
    d = json.loads(line[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 2048
    assert d['shard_check']['sharded_equals_unsharded'] is True and d['value'] > 0


@pytest.mark.parametrize('index,extra', [
    ('ivfpq', []), ('ivfflat', ['--head-keys', '600', '--shard-keys', '768', '--extras-per-query', '512']),
    ('ivfpq', ['--head-keys', '700'])])
def test_two_rank_sharded_bench_path_rccl_form_of_the_collectives(index, extra):
    """The same two ranks on GPU 0, but the collectives in the form the RCCL runs use: all_to_all_single
    / all_gather_into_tensor on the DEVICE tensors as they are, asynchronous handles, four pieces per
    batch (gloo carries device tensors for these; RCCL itself refuses two ranks on one GPU:
    scripts/probe_collectives.py)."""
    env = dict(os.environ, ASL_BENCH_BACKEND='gloo', ASL_DIRECT_COLLECTIVES='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(29611 + len(extra) + (3 if index == 'ivfflat' else 0)),
           os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '2', '--warmup', '1', '--library-size', '60000', '--nlist',
           '256', '--niter', '4', '--batch', '1024', '--recall-queries', '64', '--index', index] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert line, out.stderr[-2000:]
    d = json.loads(line[-1])
    assert d['n_gpus'] == 2 and d['shard_check']['sharded_equals_unsharded'] is True
    c = d['comm']
    assert c['fallbacks_to_full_exchange'] == 0 and c['total_bytes_out_per_rank_per_step'] > 0
    if '--shard-keys' in extra:
        assert c['exchange'] == 'two-phase, shard-side k + second scans'
    elif extra:
        assert c['exchange'].startswith('two-phase') and 'held_back_keys_all_to_all' in c


@pytest.mark.parametrize('index', ['ivfpq', 'ivfflat', 'ivfpq+rccl-form'])
def test_two_rank_sharded_cascade(index):
    """configs[4] through the real engine on two ranks sharing GPU 0 (host-side collectives):
    standard search data-parallel over the queries, open search over the list-sharded index;
    the identifications must equal the one-GPU cascade's (bench.py --workload cascade)."""
    env = dict(os.environ, ASL_BENCH_BACKEND='gloo')
    if index.endswith('+rccl-form'):       # the collectives as the RCCL runs issue them (see above)
        env['ASL_DIRECT_COLLECTIVES'] = '1'
        index = 'ivfpq'
        env['MASTER_PORT_SHIFT'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port',
           str((29541 if index == 'ivfpq' else 29542) + 2 * int(env.get('MASTER_PORT_SHIFT', 0))),
           os.path.join(ROOT, 'bench.py'), '--workload', 'cascade',
           '--gpus', '2', '--steps', '1', '--warmup', '1', '--library-size', '60000', '--nlist',
           '256', '--niter', '4', '--batch', '1024', '--cascade-batches', '2', '--index', index]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert line, out.stderr[-3000:]
    d = json.loads(line[-1])
    assert d['n_gpus'] == 2 and d['config']['workload'].startswith('configs[4]')
    assert d['cascade_check']['sharded_cascade_equals_unsharded'] is True
    lv = d['levels']
    assert lv['std']['queries_in_per_step'] == 4096
    assert 0 < lv['open']['queries_in_per_step'] < 4096              # only the remainder
    ident = d['identifications']
    assert ident['source_spectrum_identified'] > 0.6 * ident['queries']
    assert d['value'] > 0


def test_sharded_exchange_falls_back_without_packed_keys():
    """packed-key rows exist only in the tiled m = 32 / 8-bit scan; every other
    IVF-PQ shape must exchange (D, I) rows instead of failing."""
    import torch
    from ann_solo_amd import synthetic
    from ann_solo_amd.distributed import HipShardBackend
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(5000, seed=79, device='cpu', charges=(2,), charge_p=(1.0,))
    for kw, want in ((dict(pq_m=32), True), (dict(pq_m=16), False), (dict(pq_m=32, pq_bits=6), False)):
        sl = SpectralLibrary(lib, config=Config.open_search(num_list=32, num_probe=8, num_candidates=200,
                                                index='ivfpq', kmeans_niter=3, **kw))
        be = HipShardBackend(sl, 2, 'open')
        assert be.supports_keys is want, kw
        sl.shutdown()
    sl = SpectralLibrary(lib, config=Config.open_search(num_list=32, num_probe=8, num_candidates=1500,
                                            index='ivfpq', kmeans_niter=3))
    assert HipShardBackend(sl, 2, 'open').supports_keys is False      # k + 768 > 2048


_RCCL_WORLD1 = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
from ann_solo_amd.distributed import (HipShardBackend, exchange_keys, exchange_partials,
                                      _all_gather_rows, sharded_search_batch)
dev = torch.device('cuda', 0)
# the collectives exactly as the sharded search issues them (RCCL, its own stream, async)
K = torch.arange(6 * 5, dtype=torch.int64, device=dev).view(6, 5)
Ko, works, keep = exchange_keys(K, 1)
for w in works: w.wait()
assert torch.equal(Ko.view(6, 5), K)
D = torch.rand(6, 5, device=dev); I = torch.arange(30, device=dev).view(6, 5)
Do, Io, works, keep = exchange_partials(D, I, 1, async_op=True)
for w in works: w.wait()
assert torch.equal(Do.view(6, 5), D) and torch.equal(Io.view(6, 5), I)
g, w = _all_gather_rows(D, 1, async_op=True); w.wait()
assert torch.equal(g, D)
# the whole chunked pipeline over RCCL at world 1 == the unsharded search
lib, aux = synthetic.make_library(20000, seed=5, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=64, num_probe=16, num_candidates=256, index=sys.argv[3], pq_m=32,
             kmeans_niter=4, mode='ann', batch_size=512)
sl = SpectralLibrary(lib, config=cfg, device=dev)
q, _ = synthetic.make_queries(lib, aux, 512, seed=6, open_range=300.0, charge=2)
ref = sl._search_batch(q, 2, 'open', device_out=True)
sl._get_ann_index(2).shard(0, 1)
be = HipShardBackend(sl, 2, 'open')
got = sharded_search_batch(be, q, device_out=True, _force_exchange=True, check_sizes=True)
assert torch.equal(got.best_row, ref.best_row) and torch.equal(got.best_score, ref.best_score)
if sys.argv[3] == 'ivfpq':      # a PQ shape without packed keys takes the (D, I) exchange
    cfg16 = Config.open_search(num_list=64, num_probe=16, num_candidates=256, index='ivfpq', pq_m=16,
                   kmeans_niter=4, mode='ann', batch_size=512)
    s16 = SpectralLibrary(lib, config=cfg16, device=dev)
    ref16 = s16._search_batch(q, 2, 'open', device_out=True)
    s16._get_ann_index(2).shard(0, 1)
    be16 = HipShardBackend(s16, 2, 'open')
    assert not be16.supports_keys
    got16 = sharded_search_batch(be16, q, device_out=True, _force_exchange=True)
    assert torch.equal(got16.best_row, ref16.best_row) and torch.equal(got16.best_score, ref16.best_score)
    # the cascade entry point at world 1: ragged batch, both levels
    from ann_solo_amd.distributed import sharded_cascade_batch
    q37 = q.select(torch.arange(37, device=q.device))
    for mode, ann in (('open', True), ('std', False)):
        a = sharded_cascade_batch(be, q37, mode, ann)
        sl2 = sl._search_batch_local(q37, 2, mode) if mode == 'std' else None
        if sl2 is not None:
            assert (a.best_row == sl2.best_row).all() and (a.best_score == sl2.best_score).all()
        else:
            assert (a.best_row == ref.best_row[:37].cpu().numpy()).all()
dist.destroy_process_group()
print('rccl-world1-ok')
'''


@pytest.mark.parametrize('index', ['ivfpq', 'ivfflat'])
def test_rccl_collectives_at_world_one(index, tmp_path):
    """The RCCL code path itself (all_to_all_single / all_gather_into_tensor on device
    tensors, async handles, the chunked overlap) cannot be run with two ranks on one GPU; at
    world size 1 every call is still issued through RCCL exactly as in a multi-GPU job."""
    script = tmp_path / 'rccl1.py'
    script.write_text(_RCCL_WORLD1)
    out = subprocess.run([sys.executable, str(script), ROOT, '29531' if index == 'ivfpq' else '29532',
                          index], capture_output=True, text=True, timeout=600)
    assert 'rccl-world1-ok' in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


_PLANES_WORLD1 = r'''
import os, sys
from argparse import Namespace
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], RANK='0', WORLD_SIZE='1')
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('gloo')                                     # control plane, as bench.py sets it up
try:
    g = dist.new_group([0], backend='nccl', device_id=dev)          # data plane
except TypeError:
    g = dist.new_group([0], backend='nccl')
assert dist.get_backend() == 'gloo' and dist.get_backend(g) == 'nccl'
import bench
bench.preflight(Namespace(preflight_seconds=120.0), 1, 0, dev, 'nccl', g)
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
from ann_solo_amd.distributed import HipShardBackend, make_shard_groups, sharded_search_batch
group, srank, gi = make_shard_groups(1, backend='nccl', world_group=g)
assert group is g and srank == 0 and gi == 0
lib, aux = synthetic.make_library(20000, seed=5, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=64, num_probe=16, num_candidates=256, index='ivfpq', pq_m=32,
             kmeans_niter=4, mode='ann', batch_size=512)
sl = SpectralLibrary(lib, config=cfg, device=dev)
q, _ = synthetic.make_queries(lib, aux, 512, seed=6, open_range=300.0, charge=2)
ref = sl._search_batch(q, 2, 'open', device_out=True)
sl._get_ann_index(2).shard(0, 1)
be = HipShardBackend(sl, 2, 'open')
got = sharded_search_batch(be, q, group=group, device_out=True, _force_exchange=True, check_sizes=True)
assert torch.equal(got.best_row, ref.best_row) and torch.equal(got.best_score, ref.best_score)
t = torch.tensor([3.5], dtype=torch.float64)                       # the clock's reduction: gloo, host tensor
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
assert float(t) == 3.5
dist.destroy_process_group()
print('planes-world1-ok')
'''


def test_gloo_control_plane_with_an_rccl_data_plane_at_world_one(tmp_path):
    """bench.py at N > 1 keeps barriers / the clock / flags on a gloo default group and creates the
    RCCL group of the sharded search's collectives only after the replicas line is in reserve. The
    same set-up at world 1: default group gloo, `new_group(backend='nccl')`, bench's own preflight over
    it, `make_shard_groups(..., backend='nccl', world_group=...)`, a sharded batch on that group."""
    script = tmp_path / 'planes1.py'
    script.write_text(_PLANES_WORLD1)
    out = subprocess.run([sys.executable, str(script), ROOT, '29541'], capture_output=True, text=True, timeout=600)
    assert 'planes-world1-ok' in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


_RCCL_CABI = r'''
import ctypes as C, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
torch.cuda.set_device(0)
# the communicator comes from the host application -- here: the RCCL copy PyTorch ships, made
# visible process-wide so that libannsolo_mi resolves the very same instance
rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'), mode=C.RTLD_GLOBAL)
class UID(C.Structure):
    _fields_ = [('b', C.c_char * 128)]
uid, comm = UID(), C.c_void_p()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
from ann_solo_amd import _lib, synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(20000, seed=5, device=dev, charges=(2,), charge_p=(1.0,))
q, _ = synthetic.make_queries(lib, aux, 300, seed=6, open_range=300.0, charge=2)
L = _lib.lib()
for index in ('ivfpq', 'ivfflat', 'ivfpq+refine'):
    # ('ivfpq+refine': the exact re-rank on -- the shards' un-refined k' rows travel whole, the owner
    # re-ranks the merged short-list: the (D, I) exchange inside the library)
    sl = SpectralLibrary(lib, config=Config.open_search(num_list=64, num_probe=16, num_candidates=256,
                                            index=index.split('+')[0], kmeans_niter=4,
                                            refine_k=400 if index.endswith('refine') else None), device=dev)
    idx = sl._get_ann_index(2)
    vec = sl._encode(q)
    idx.nprobe = 16
    D0, I0 = idx.search(vec, 256)
    # unsharded handles are refused, then the real thing at world 1
    D = torch.empty((q.n, 256), dtype=torch.float32, device=dev)
    I = torch.empty((q.n, 256), dtype=torch.int64, device=dev)
    idx.shard(0, 1)
    rc = L.asl_index_search_sharded(idx._h, comm, q.n, _lib.ptr(vec), 256, 16, _lib.ptr(D), _lib.ptr(I))
    assert rc == 0, L.asl_last_error()
    torch.cuda.synchronize()
    assert torch.equal(I, I0) and torch.equal(D, D0), index
    # every step of the exact exchange inside the library (at world 1 the default head is the whole
    # row: the sizes are given): heads of 100 keys + bound + held-back keys; shards with their own
    # k = 160 and a second, full-k scan of the rows the bound asks about (few queries: all fit);
    # no room for the answers / for the second scans: the batch falls back to the full rows
    for nq_, head, sk, xper in ((q.n, 100, 0, -1), (q.n, 100, 0, 256), (40, 100, 160, 256), (q.n, 100, 160, 256),
                                (q.n, 100, 0, 0), (q.n, 255, 0, -1), (40, 8, 64, 256)):
        D.fill_(-5.0)
        I.fill_(-5)
        rc = L.asl_index_search_sharded_ex(idx._h, comm, nq_, _lib.ptr(vec), 256, 16, _lib.ptr(D), _lib.ptr(I),
                                           head, sk, xper)
        assert rc == 0, (L.asl_last_error(), head, sk, xper)
        torch.cuda.synchronize()
        # (the exchange returns sets: rows ordered only by the final merge -- compare as sorted rows)
        assert torch.equal(I[:nq_].sort(1).values, I0[:nq_].sort(1).values), (index, nq_, head, sk, xper)
        assert torch.equal(D[:nq_].sort(1).values, D0[:nq_].sort(1).values), (index, nq_, head, sk, xper)
        assert torch.equal(I[:nq_], I0[:nq_]), (index, nq_, head, sk, xper)
    # the queries travel as entry lists (<= 64 non-zero components); a query with more has no such
    # form: the ranks notice in the step's agreement and repeat the batch with dense rows
    vec2 = vec.clone()
    g = torch.Generator(device='cpu').manual_seed(3)
    for r_ in (5, 200):
        dense = torch.rand(vec.shape[1], generator=g).to(dev) * (torch.rand(vec.shape[1], generator=g).to(dev) < 0.2)
        vec2[r_] = dense / dense.norm()
    assert int((vec2[5] != 0).sum()) > 64
    D2, I2 = idx.search(vec2, 256)      # (sharding over one rank leaves the index as it is)
    for head, sk, xper in ((0, 0, -1), (100, 160, 256), (100, 0, 0)):
        D.fill_(-5.0)
        I.fill_(-5)
        rc = L.asl_index_search_sharded_ex(idx._h, comm, q.n, _lib.ptr(vec2), 256, 16, _lib.ptr(D), _lib.ptr(I),
                                           head, sk, xper)
        assert rc == 0, (L.asl_last_error(), head, sk, xper)
        torch.cuda.synchronize()
        assert torch.equal(I, I2) and torch.equal(D, D2), (index, 'dense query', head, sk, xper)
    # host pointers and missing communicators are errors, not crashes
    assert L.asl_index_search_sharded(idx._h, None, q.n, _lib.ptr(vec), 256, 16, _lib.ptr(D), _lib.ptr(I)) < 0
    host = np.zeros((q.n, 256), np.int64)
    assert L.asl_index_search_sharded(idx._h, comm, q.n, _lib.ptr(vec), 256, 16, _lib.ptr(D), _lib.ptr(host)) < 0
    sl.shutdown()
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print('rccl-cabi-ok')
'''


def test_c_abi_sharded_search_over_a_caller_owned_rccl_communicator(tmp_path):
    """asl_index_search_sharded (the rcclComm form of the boundary): RCCL
    all-gathers + grouped send/recv issued from inside the library on a communicator the host
    created; at world 1 every collective still goes through RCCL and the result must equal the
    unsharded search bit for bit."""
    script = tmp_path / 'rccl_cabi.py'
    script.write_text(_RCCL_CABI)
    out = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert 'rccl-cabi-ok' in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def test_search_preassigned_equals_search():
    import torch
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(5000, seed=77, device='cpu', charges=(2,), charge_p=(1.0,))
    sl = SpectralLibrary(lib, config=Config.open_search(num_list=32, num_probe=8, num_candidates=200,
                                            index='ivfpq', kmeans_niter=4))
    q, _ = synthetic.make_queries(lib, aux, 100, seed=78, charge=2)
    idx = sl._get_ann_index(2)
    vec = sl._encode(q.to('cuda'))
    D, I = idx.search(vec, 200)
    cD, cI = idx.coarse(vec, 8)
    D2, I2 = idx.search_preassigned(vec, 200, cD, cI)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    # dropping probes (-1) only removes candidates of those lists
    cI3 = cI.clone()
    cI3[:, 4:] = -1
    idx.nprobe = 4
    D4, I4 = idx.search(vec, 200)
    D3, I3 = idx.search_preassigned(vec, 200, cD, cI3)
    assert torch.equal(I3, I4)
    # the same contract for IVF-Flat (both scan formulations)
    sf = SpectralLibrary(lib, config=Config.open_search(num_list=32, num_probe=8, num_candidates=200,
                                            index='ivfflat', kmeans_niter=4))
    fidx = sf._get_ann_index(2)
    for variant in (0, 1):
        fidx.set_scan_variant(variant)
        fidx.nprobe = 8
        Df, If = fidx.search(vec, 200)
        fD, fI = fidx.coarse(vec, 8)
        Df2, If2 = fidx.search_preassigned(vec, 200, fD, fI)
        assert torch.equal(If, If2) and torch.equal(Df, Df2), variant
        fI3 = fI.clone()
        fI3[:, 4:] = -1
        fidx.nprobe = 4
        _, If4 = fidx.search(vec, 200)
        _, If3 = fidx.search_preassigned(vec, 200, fD, fI3)
        assert torch.equal(If3, If4), variant
    fidx.set_scan_variant(0)

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


@pytest.mark.parametrize('degree', [0, 1])
def test_two_rank_sharded_bench_path(degree):
    """degree 0 = lists sharded over both ranks; degree 1 = two replicas (no exchange)"""
    env = dict(os.environ, ASL_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(29517 + degree),
           os.path.join(ROOT, 'bench.py'),
           '--gpus', '2', '--steps', '1', '--warmup', '1', '--library-size', '60000', '--nlist',
           '256', '--niter', '4', '--batch', '1024', '--recall-queries', '64',
           '--shard-degree', str(degree)]
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
    else:
        assert d['config']['parallelism'] == 'replicas x2' and d['alt_layouts'] is None


def test_search_preassigned_equals_search():
    import torch
    from ann_solo_amd import synthetic
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(5000, seed=77, device='cpu', charges=(2,), charge_p=(1.0,))
    sl = SpectralLibrary(lib, config=Config(num_list=32, num_probe=8, num_candidates=200,
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

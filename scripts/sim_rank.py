"""Measurement helper: the compute one rank of a W-GPU sharded search performs per step,
emulated on ONE GPU (shard 0 of W, W x batch queries). Collectives are not included.

  python scripts/sim_rank.py W [library_size] [batch] [scan_variant] [ivfpq|ivfflat]
"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
from ann_solo_amd.distributed import HipShardBackend

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_100_000
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
index = sys.argv[5] if len(sys.argv) > 5 else 'ivfpq'
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(n, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config(num_list=4096, num_probe=128, num_candidates=1024, index=index, pq_m=32,
             kmeans_niter=25, mode='ann', precursor_tolerance_mass_open=500.0,
             precursor_tolerance_mode_open='Da', batch_size=batch, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
idx.set_scan_variant(variant)
q_all, _ = synthetic.make_queries(lib, aux, W * batch, seed=42, open_range=500.0, charge=2)
q = q_all.select(torch.arange(batch, device=dev)).contiguous()
be = HipShardBackend(sl, 2, 'open')
allvec = be.encode(q_all)
cD, cI = be.coarse(allvec)
idx.shard(0, W)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


t_enc, _ = timed(lambda: be.encode(q_all))      # every rank hashes ALL queries (peaks travel)
vec = be.encode(q)
t_coarse, _ = timed(lambda: be.coarse(vec))     # ... and quantises its own slice
if index == 'ivfpq':
    t_scan, K = timed(lambda: be.shard_search_keys(allvec, cD, cI))
    Ks = K.view(W, batch, -1).contiguous()
    t_merge, (_, knn) = timed(lambda: be.merge_keys(Ks))
else:
    t_scan, (D, I) = timed(lambda: be.shard_search_preassigned(allvec, cD, cI))
    Ds, Is = D.view(W, batch, -1).contiguous(), I.view(W, batch, -1).contiguous()
    t_merge, (_, knn) = timed(lambda: be.merge(Ds, Is))
t_resc, _ = timed(lambda: be.rescore_knn(q, knn, True))
tot = t_enc + t_coarse + t_scan + t_merge + t_resc
print(f'{index} W={W} batch/rank={batch} variant={variant}: encode (all {W * batch}) {t_enc:.2f} coarse {t_coarse:.2f} '
      f'shard scan ({W * batch} queries) {t_scan:.2f} merge {t_merge:.2f} rescore {t_resc:.2f} '
      f'| compute per step {tot:.2f} ms (collectives excluded)')

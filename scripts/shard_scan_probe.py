"""Measurement helper: what the shard scan of one rank of a W-GPU search spends its time on.
Shard 0 of W, W x batch queries (as scripts/sim_rank.py), timed with the hit count k and the
number of probed lists varied -- the intercept at zero lists is the per-(query, shard) fixed cost,
the dependence on k the cost of maintaining the per-shard top-k.

  python scripts/shard_scan_probe.py W [ivfpq|ivfflat] [library_size] [batch] [quick]
"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
from ann_solo_amd.distributed import HipShardBackend

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
index = sys.argv[2] if len(sys.argv) > 2 else 'ivfpq'
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2_100_000
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 16384
quick = len(sys.argv) > 5 and sys.argv[5] == 'quick'      # k 1024 with 128 / 8 probes only
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(n, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=128, num_candidates=1024, index=index, pq_m=32,
             kmeans_niter=25, mode='ann', precursor_tolerance_mass_open=500.0,
             precursor_tolerance_mode_open='Da', batch_size=batch, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
q_all, _ = synthetic.make_queries(lib, aux, W * batch, seed=42, open_range=500.0, charge=2)
be = HipShardBackend(sl, 2, 'open')
allvec = be.encode(q_all)
cD, cI = be.coarse(allvec)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


t1 = timed(lambda: idx.search_preassigned_keys(allvec[:batch], 1024, cD[:batch], cI[:batch]))
print(f'{index} unsharded, {batch} queries, nprobe 128, k 1024: {t1:.2f} ms', flush=True)
if W > 1:
    idx.shard(0, W)
for k in ((1024,) if quick else (1024, 512, 256, 64)):
    for npr in ((128, 8) if quick else (128, 64, 32, 8)):
        d_, i_ = cD[:, :npr].contiguous(), cI[:, :npr].contiguous()
        t = timed(lambda: idx.search_preassigned_keys(allvec, k, d_, i_))
        print(f'{index} shard 0 of {W}, {W * batch} queries, first {npr} probes (~{npr / W:.0f} lists here), k {k}: {t:.2f} ms',
              flush=True)

"""Measurement helper: how many probed lists do two queries of a batch share? (the reuse a
workgroup scanning two queries' lists together could win)   python scripts/probe_overlap.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary

dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=128, num_candidates=1024, index='ivfpq', pq_m=32,
             kmeans_niter=10, mode='ann', batch_size=16384, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
vec = sl._encode(q)
cD, cI = idx.coarse(vec, 128)
n = vec.shape[0]
sizes = torch.as_tensor(idx.list_sizes(), device=dev).float() if hasattr(idx, 'list_sizes') else None
B = torch.zeros((n, 4096), dtype=torch.float16, device=dev)
B.scatter_(1, cI.long().clamp_min(0), 1.0)
best = torch.zeros(n, device=dev)
for lo in range(0, n, 2048):
    O = (B[lo:lo + 2048] @ B.T).float()
    O[torch.arange(O.shape[0], device=dev), torch.arange(lo, lo + O.shape[0], device=dev)] = -1
    best[lo:lo + 2048] = O.max(1).values
print('shared probes with the BEST partner in the batch: mean %.1f of 128 (p10 %.0f, p50 %.0f, p90 %.0f)'
      % (best.mean(), best.quantile(0.1), best.quantile(0.5), best.quantile(0.9)))
key = cI[:, 0].long() * 4096 + cI[:, 1].long()
order = key.argsort()
Bs = B[order]
adj = (Bs[0::2] * Bs[1::2]).sum(1).float()
print('pairs of neighbours after sorting by the two top probes: mean %.1f shared' % adj.mean())
# greedy matching: visit queries in random order, take the best still-free partner
free = torch.ones(n, dtype=torch.bool, device=dev)
tot, cnt = 0.0, 0
perm = torch.randperm(n, device=dev)[:2048]
for i in perm.tolist():
    if not free[i]:
        continue
    free[i] = False
    o = (B @ B[i]).float()
    o[~free] = -1
    j = int(o.argmax())
    free[j] = False
    tot += float(o[j]); cnt += 1
print('greedy matching (first %d pairs): mean %.1f shared' % (cnt, tot / cnt))

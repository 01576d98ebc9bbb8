"""Measurement helper: does the order of the queries in a batch (locality of the probed lists
among concurrently running workgroups) change the PQ scan time?  python scripts/query_order.py"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary

dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=128, num_candidates=1024, index='ivfpq', pq_m=32,
             kmeans_niter=25, mode='ann', batch_size=16384, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
idx.nprobe = 128
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
vec = sl._encode(q)
cD, cI = idx.coarse(vec, 128)
idx.set_unordered(True)


def timed(v, d, i, reps=5):
    idx.search_preassigned(v, 1024, d, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        idx.search_preassigned(v, 1024, d, i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n = vec.shape[0]
print('batch order         : %.2f ms' % timed(vec, cD, cI))
key = cI[:, 0].long() * 4096 + cI[:, 1].long()
order = key.argsort()
print('sorted by top probes: %.2f ms' % timed(vec[order].contiguous(), cD[order].contiguous(), cI[order].contiguous()))
j = torch.arange(n, device=dev)
blk = (j % (n // 8)) * 8 + j // (n // 8)       # sorted position j runs as workgroup blk[j]: XCD = j / (n/8)
perm = torch.empty_like(order)
perm[blk] = order
print('sorted, one XCD per sorted range: %.2f ms' % timed(vec[perm].contiguous(), cD[perm].contiguous(), cI[perm].contiguous()))
# canonical probe order: every workgroup walks its lists by ascending list id, so workgroups that
# run at the same time and share lists read them at about the same moment (L2 hits?)
sI, so = cI.sort(dim=1)
sD = cD.gather(1, so)
print('lists by id, batch order            : %.2f ms' % timed(vec, sD.contiguous(), sI.contiguous()))
print('lists by id, sorted by top probes   : %.2f ms' % timed(vec[order].contiguous(), sD[order].contiguous(), sI[order].contiguous()))
print('lists by id, sorted, XCD ranges     : %.2f ms' % timed(vec[perm].contiguous(), sD[perm].contiguous(), sI[perm].contiguous()))
# similar queries next to each other: order by the centroid the query is closest to, then by score
key2 = cI[:, 0].long() * 1000000 + (cD[:, 0] * 999999).long().clamp(0, 999999)
o2 = key2.argsort()
p2 = torch.empty_like(o2)
p2[blk] = o2
print('lists by id, by top-1 list+score, XCD: %.2f ms' % timed(vec[p2].contiguous(), sD[p2].contiguous(), sI[p2].contiguous()))
rnd = torch.randperm(n, device=dev)
print('random order        : %.2f ms' % timed(vec[rnd].contiguous(), cD[rnd].contiguous(), cI[rnd].contiguous()))

# longest-first: workgroups are dispatched in blockIdx order, so the short queries fill the tail
offs = idx.lists()[0]
sizes = torch.as_tensor(offs[1:] - offs[:-1], device=dev).float()
load = sizes[cI.long().clamp_min(0)].sum(1)
lpt = load.argsort(descending=True)
print('longest first       : %.2f ms' % timed(vec[lpt].contiguous(), cD[lpt].contiguous(), cI[lpt].contiguous()))
spt = load.argsort()
print('shortest first      : %.2f ms' % timed(vec[spt].contiguous(), cD[spt].contiguous(), cI[spt].contiguous()))
print('batch order again   : %.2f ms' % timed(vec, cD, cI))
print('load: mean %.0f  min %.0f  max %.0f vectors/query' % (load.mean(), load.min(), load.max()))

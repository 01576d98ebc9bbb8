"""Measurement helper: how many shards would a query touch if list ownership followed the
geometry of the centroids (k-means of the centroids into W groups) instead of load-only LPT?
   python scripts/probe_locality.py [W]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config.open_search(num_list=4096, num_probe=128, num_candidates=1024, index='ivfpq', pq_m=32,
             kmeans_niter=25, mode='ann', batch_size=16384, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
vec = sl._encode(q)
_, cI = idx.coarse(vec, 128)
cI = cI.long()
off = torch.as_tensor(idx.lists()[0].astype(np.int64), device=dev)
sizes = (off[1:] - off[:-1]).float()
cen = torch.as_tensor(idx.centroids(), device=dev)


def report(tag, owner):
    own = owner[cI]                                           # [nq, nprobe]
    load = sizes[cI]
    per = torch.stack([(load * (own == r)).sum(1) for r in range(W)], 1)   # scanned vectors per (q, shard)
    touched = (per > 0).sum(1).float()
    srt = per.sort(1, descending=True)[0]
    frac = srt.cumsum(1) / srt.sum(1, keepdim=True)
    shard_load = per.mean(0)
    print(f'{tag}: shards touched/query mean {touched.mean():.2f} (min {touched.min():.0f} max {touched.max():.0f}); '
          f'vectors in the top 1/2/3 shards {frac[:, 0].mean():.2f}/{frac[:, 1].mean():.2f}/{frac[:, 2].mean():.2f}; '
          f'shard load max/mean {shard_load.max() / shard_load.mean():.3f}')


owner_lpt = torch.as_tensor(idx.shard_map(W).astype(np.int64), device=dev)
report('load-balanced LPT (current)', owner_lpt)
# spherical k-means of the centroids into W groups
g = torch.Generator(device='cpu').manual_seed(1)
cn = torch.nn.functional.normalize(cen, dim=1)
mu = cn[torch.randperm(cn.shape[0], generator=g)[:W].to(dev)].clone()
for _ in range(30):
    a = (cn @ mu.T).argmax(1)
    for r in range(W):
        if (a == r).any():
            mu[r] = torch.nn.functional.normalize(cn[a == r].mean(0), dim=0)
report('k-means of centroids (unbalanced)', a)
# balanced variant: fill groups greedily by affinity under a load cap (size^2 weights)
w = sizes * sizes
cap = w.sum() / W * 1.02
aff = cn @ mu.T
order = (aff.max(1)[0] - aff.topk(2, 1)[0][:, 1]).argsort(descending=True)   # most decided first
owner_b = torch.full((cn.shape[0],), -1, dtype=torch.long, device=dev)
fill = torch.zeros(W, device=dev)
aff_c, w_c, order_c = aff.cpu().numpy(), w.cpu().numpy(), order.cpu().numpy()
ob, fl = np.full(cn.shape[0], -1), np.zeros(W)
for i in order_c:
    for r in np.argsort(-aff_c[i]):
        if fl[r] + w_c[i] <= float(cap) or r == np.argsort(-aff_c[i])[-1]:
            ob[i] = r
            fl[r] += w_c[i]
            break
report('k-means of centroids, load-capped', torch.as_tensor(ob, device=dev))

"""Measurement helper: probe-weighted scan load of every shard under the list ownership
of asl_lpt_owner.   python scripts/shard_balance.py [W]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary
from ann_solo_amd.distributed import lpt_owner

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
cI = cI.cpu().numpy()
off = idx.lists()[0].astype(np.int64)
sizes = off[1:] - off[:-1]
owner = idx.shard_map(W)
assert (owner == lpt_owner(sizes * sizes, W)).all()
tiles = (sizes + 63) // 64
print('list sizes: min %d mean %.0f max %d; tiles/list mean %.2f (ideal %.2f)' %
      (sizes.min(), sizes.mean(), sizes.max(), tiles.mean(), sizes.mean() / 64))


def report(tag, owner):
    vec_load = np.zeros(W)
    tile_load = np.zeros(W)
    own = owner[cI]                       # [nq, nprobe]
    for r in range(W):
        sel = own == r
        vec_load[r] = (sizes[cI] * sel).sum() / cI.shape[0]
        tile_load[r] = (tiles[cI] * sel).sum() / cI.shape[0]
    print(tag, 'owned vectors per shard:', [int(sizes[owner == r].sum()) for r in range(W)])
    print(tag, 'scanned vectors/query per shard:', np.round(vec_load).astype(int).tolist(),
          'max/mean %.3f' % (vec_load.max() / vec_load.mean()))
    print(tag, 'tiles/query per shard:', np.round(tile_load).astype(int).tolist(),
          'max/mean %.3f' % (tile_load.max() / tile_load.mean()))


report('[lpt len]', lpt_owner(sizes, W))
report('[lpt len^2 = asl_index_shard]', owner)
freq = np.bincount(cI.reshape(-1), minlength=len(sizes)).astype(np.int64)
report('[lpt len*freq (oracle of the batch)]', lpt_owner(sizes * freq, W))
print('corr(len, probe freq) = %.3f' % np.corrcoef(sizes, freq)[0, 1])

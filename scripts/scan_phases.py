"""Measurement helper: per-workgroup phase durations of pq_scan_v3 (dbg bit 32), for the
unsharded index and for shard 0 of W. Needs the instrumented library
(make -C ann_solo_amd/csrc clean all EXTRA=-DASL_ENABLE_DBG).   python scripts/scan_phases.py [W] [variant]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from ann_solo_amd import synthetic
from ann_solo_amd.spectral_library import Config, SpectralLibrary

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda', 0)
lib, aux = synthetic.make_library(2_100_000, seed=20240807, device=dev, charges=(2,), charge_p=(1.0,))
cfg = Config(num_list=4096, num_probe=128, num_candidates=1024, index='ivfpq', pq_m=32,
             kmeans_niter=10, mode='ann', batch_size=16384, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
q, _ = synthetic.make_queries(lib, aux, 16384, seed=42, open_range=500.0, charge=2)
vec = sl._encode(q)
idx.nprobe = 128


def phases(tag):
    idx.set_scan_variant(variant | (32 << 8))
    D, _ = idx.search(vec, 1024)
    torch.cuda.synchronize()
    D = D[:, :7].double().cpu()
    m = D.mean(0)
    us = m[:4] / 100.0
    print(f'{tag}: LUT {us[0]:.1f} us | init {us[1]:.1f} | tile loop {us[2]:.1f} | finish {us[3]:.1f} '
          f'(compaction {m[5] / 100:.1f}, ids+sort {m[6] / 100:.1f}) | tiles/query {m[4]:.0f} | sum {us.sum():.1f} us/workgroup')
    idx.set_scan_variant(variant)


phases('unsharded')
idx.shard(0, W)
phases(f'shard 0 of {W}')

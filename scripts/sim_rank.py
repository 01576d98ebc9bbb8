"""Measurement helper: the compute one rank of a W-GPU sharded search performs per step,
emulated on ONE GPU (shard 0 of W, W x batch queries). Collectives are not included.

  python scripts/sim_rank.py W [library_size] [batch] [scan_variant] [ivfpq|ivfflat] [shard_k (timing only)]
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
if len(sys.argv) > 6:          # what-if: a shard-side k smaller than k (results are NOT exact then: timing only)
    be.k_scan = int(sys.argv[6])
allvec = be.encode(q_all)
cD, cI = be.coarse(allvec)
# the W shards of the index, side by side on this GPU (copies through a file): the rows every shard
# holds for this rank's own `batch` queries are what the exchange statistics need
import tempfile
from ann_solo_amd import faiss_compat as faiss
tmp = os.path.join(tempfile.mkdtemp(), 'sim.idxmi')
faiss.write_index(idx, tmp)
own_rows = []
for s_ in range(1, W):
    other = faiss.read_index(tmp)
    other.shard(s_, W)
    own_rows.append(other.search_preassigned_keys(allvec[:batch], be.k_scan, cD[:batch], cI[:batch]))
    del other
os.remove(tmp)
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
from ann_solo_amd.distributed import head_width
# the two-phase exchange's compute on this rank: split of the W x batch local rows, merge of the
# heads this rank would own (here: the heads of its own rows for the first `batch` queries of
# every "shard" -- the same work), the held-back keys above the bounds, the final merge
t_scan, K = timed(lambda: be.shard_search_keys(allvec, cD, cI))
kp = head_width(be.k_scan, W)
t_split, (head, floor) = timed(lambda: be.keys_split(K, kp))
# what this rank receives: the heads of its own queries' rows from every shard
heads = torch.stack([head[:batch]] + [be.keys_split(r_, kp)[0] for r_ in own_rows]).contiguous()
rows_own = [K[:batch].contiguous()] + own_rows          # every shard's rows for this rank's own queries ...
floors = [floor[:batch].contiguous()] + [be.keys_split(r_, kp)[1] for r_ in own_rows]     # ... and their floors
t_m1, (okeys, bnd, need) = timed(lambda: be.keys_merge_heads(heads, be.k_scan))
t_x = t_m2 = 0.0
if kp - 1 < be.k_scan:
    flag = be.new_flag()
    xcap = batch * max(8, be.k_scan // 16)
    # shard side: this rank answers W x batch rows (timed on its own rows against the bounds its
    # own queries got -- the same amount of work); owner side: the real answers of the W shards
    t_x, _ = timed(lambda: be.keys_extras(K, floor, bnd.reshape(-1).contiguous(), W, xcap, be.new_flag()))
    xbuf = torch.stack([be.keys_extras(rows_own[s_], floors[s_], bnd[s_].contiguous(), 1, xcap, flag)[0]
                        for s_ in range(W)])
    t_m2, knn = timed(lambda: be.keys_merge_final(heads, xbuf, okeys, need, be.k_scan))
    want = be.merge_keys(torch.stack([K[:batch]] + own_rows).contiguous())[1]
    exact = bool(torch.equal(torch.sort(knn, 1).values, torch.sort(want, 1).values))
    held = float(((xbuf[:, :batch] >> 32).sum()).item()) / batch
    asked = float(need.float().mean())
else:
    t_m2, knn = timed(lambda: be.keys_merge_final(heads, None, okeys, need, be.k_scan))
    flag, asked, exact, held = None, 0.0, True, 0.0
t_old, _ = timed(lambda: be.merge_keys(K.view(W, batch, -1).contiguous()))
t_resc, _ = timed(lambda: be.rescore_knn(q, knn, True))
t_merge = t_split + t_m1 + t_x + t_m2
tot = t_enc + t_coarse + t_scan + t_merge + t_resc
print(f'{index} W={W} batch/rank={batch} variant={variant}: encode (all {W * batch}) {t_enc:.2f} coarse {t_coarse:.2f} '
      f'shard scan ({W * batch} queries) {t_scan:.2f} exchange compute {t_merge:.2f} (split {t_split:.2f} + heads {t_m1:.2f} + '
      f'held-back {t_x:.2f} + final {t_m2:.2f}; head width {kp}, queries asking {asked:.3f}, held-back keys per query {held:.1f}, '
      f'equals the full merge: {exact}, overflow '
      f'{int(flag.item()) if flag is not None else 0}; the full-row merge it replaces {t_old:.2f}) rescore {t_resc:.2f} '
      f'| compute per step {tot:.2f} ms (collectives excluded)')

# how the final top-k spreads over the shards: per query the largest number of its k best hits
# that ONE shard holds (what a shard-side k smaller than k would have to cover)
rows = torch.stack([K[:batch]] + own_rows)                      # [W, batch, k] packed keys
flip = torch.tensor(-2 ** 63, dtype=torch.int64, device=dev)   # unsigned order as signed order
allk = (rows ^ flip).permute(1, 0, 2).reshape(batch, -1)
kth = torch.topk(allk, be.k_scan, dim=1).values[:, -1:]         # the k-th best key of the union
share = ((rows ^ flip) >= kth.unsqueeze(0)).sum(2)              # [W, batch]
mx = share.max(0).values.float()
qs = torch.quantile(mx, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev)).tolist()
print(f'largest share of a query\'s top {be.k_scan} in one shard: median {qs[0]:.0f}, 90 % {qs[1]:.0f}, 99 % {qs[2]:.0f}, '
      f'99.9 % {qs[3]:.0f}, max {qs[4]:.0f}; queries with a shard holding more than 512: {(mx > 512).float().mean():.5f}, '
      f'more than 384: {(mx > 384).float().mean():.5f}, more than 640: {(mx > 640).float().mean():.5f}')

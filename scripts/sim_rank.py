"""Measurement helper: the compute one rank of a W-GPU sharded search performs per step,
emulated on ONE GPU (shard 0 of W, W x batch queries). Collectives are not included.

  python scripts/sim_rank.py W [library_size] [batch] [scan_variant] [ivfpq|ivfflat] [shard_k (default asl_shard_k(k, W); k = the round-4 protocol)]
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
from ann_solo_amd.distributed import head_width, shard_k
k = be.k_scan
# the shards' own k: asl_shard_k by default (k / 2 at 8 ranks, 5 k / 8 at 4: the third phase keeps the
# result exact); argv[6] overrides it (k itself: the round-4 protocol)
k_row = int(sys.argv[6]) if len(sys.argv) > 6 else shard_k(k, W)
nkeep = head_width(k, W) - 1
if not (nkeep < k_row < k):
    k_row = k
third = k_row < k
allvec = be.encode(q_all)
cD, cI = be.coarse(allvec)
# the W shards of the index, side by side on this GPU (copies through a file): the rows every shard
# holds for this rank's own `batch` queries are what the exchange statistics need -- the k_row-deep
# rows that travel and the FULL rows a third-phase rescan would see
import tempfile
from ann_solo_amd import faiss_compat as faiss
tmp = os.path.join(tempfile.mkdtemp(), 'sim.idxmi')
faiss.write_index(idx, tmp)
own_rows, own_full = [], []
for s_ in range(1, W):
    other = faiss.read_index(tmp)
    other.shard(s_, W)
    own_full.append(other.search_preassigned_keys(allvec[:batch], k, cD[:batch], cI[:batch]))
    own_rows.append(other.search_preassigned_keys(allvec[:batch], k_row, cD[:batch], cI[:batch]) if third
                    else own_full[-1])
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
# the exchange's compute on this rank: split of the W x batch local rows, merge of the heads this
# rank would own (here: the heads of its own rows for the first `batch` queries of every "shard" --
# the same work), the held-back keys above the bounds, the final merge, and the third phase
t_scan, K = timed(lambda: be.shard_search_keys(allvec, cD, cI, k=k_row))
t_scan_k = None
if third:
    t_scan_k, _ = timed(lambda: be.shard_search_keys(allvec, cD, cI, k=k))
full0 = be.shard_search_keys(allvec[:batch], cD[:batch], cI[:batch], k=k)
kp = nkeep + 1 + int(third)
t_split, (head, floor) = timed(lambda: be.keys_split(K, kp, third))
# what this rank receives: the heads of its own queries' rows from every shard
heads = torch.stack([head[:batch]] + [be.keys_split(r_, kp, third)[0] for r_ in own_rows]).contiguous()
rows_own = [K[:batch].contiguous()] + own_rows          # every shard's rows for this rank's own queries ...
rows_full = [full0] + own_full
floors = [floor[:batch].contiguous()] + [be.keys_split(r_, kp, third)[1] for r_ in own_rows]     # ... and their floors
t_m1, (okeys, bnd, need) = timed(lambda: be.keys_merge_heads(heads, k, third))
t_x = t_m2 = t_3 = 0.0
n3 = q3 = 0
if nkeep < k_row:
    flag = be.new_flag()
    xcap = batch * max(8, k // 16)
    # shard side: this rank answers W x batch rows (timed on its own rows against the bounds its
    # own queries got -- the same amount of work); owner side: the real answers of the W shards
    t_x, _ = timed(lambda: be.keys_extras(K, floor, bnd.reshape(-1).contiguous(), W, xcap, be.new_flag()))
    xbuf = torch.stack([be.keys_extras(rows_own[s_], floors[s_], bnd[s_].contiguous(), 1, xcap, flag)[0]
                        for s_ in range(W)])
    if third:
        t_m2, (knn, fin, req, need3) = timed(lambda: be.keys_merge_final(heads, xbuf, okeys, need, k, be.new_flag()))
        # third phase. Shard side: the requests the W owners address to ONE shard are as many as one
        # owner addresses to the W shards -- timed as a scan with the full k of exactly those (query,
        # shard) rows on this shard. Owner side: the real answers from every shard's FULL rows.
        sels = [be.request_rows(req[s_].contiguous()) for s_ in range(W)]
        allsel = torch.cat(sels)
        n3 = int(allsel.numel())
        xcap3 = batch * max(8, k // 16)
        f3 = be.new_flag()

        def shard_side():
            K3 = be.shard_search_keys(allvec.index_select(0, allsel), cD.index_select(0, allsel),
                                      cI.index_select(0, allsel), k=k)
            return be.keys_rescan(K3, allsel, req.reshape(-1, 2)[:batch].contiguous(), 1, batch, xcap3 * W, be.new_flag())
        t_rescan, _ = timed(shard_side) if n3 else (0.0, None)
        ans = torch.stack([be.keys_rescan(rows_full[s_].index_select(0, sels[s_]), sels[s_], req[s_].contiguous(),
                                          1, batch, xcap3, f3)[0] for s_ in range(W)])
        t_m3, knn3 = timed(lambda: be.keys_merge3(fin, ans, need3, k))
        sel3 = torch.nonzero(need3).reshape(-1)
        q3 = int(sel3.numel())
        t_r3 = 0.0
        if q3:
            sub = q.select(sel3)
            kn = knn3.index_select(0, sel3).contiguous()
            t_r3, _ = timed(lambda: be.rescore_knn(sub, kn, True))
        t_3 = t_rescan + t_m3 + t_r3
        knn = knn3
        flag[0] = max(int(flag[0]), int(f3[0]))
    else:
        t_m2, knn = timed(lambda: be.keys_merge_final(heads, xbuf, okeys, need, k))
    want = be.merge_keys(torch.stack(rows_full).contiguous())[1]
    exact = bool(torch.equal(torch.sort(knn, 1).values, torch.sort(want, 1).values))
    held = float(((xbuf[:, :batch] >> 32).sum()).item()) / batch
    asked = float(need.float().mean())
else:
    t_m2, knn = timed(lambda: be.keys_merge_final(heads, None, okeys, need, k))
    flag, asked, exact, held = None, 0.0, True, 0.0
t_old, _ = timed(lambda: be.merge_keys(K.view(W, batch, -1).contiguous()))
t_resc, _ = timed(lambda: be.rescore_knn(q, knn, True))
t_merge = t_split + t_m1 + t_x + t_m2
tot = t_enc + t_coarse + t_scan + t_merge + t_3 + t_resc
print(f'{index} W={W} batch/rank={batch} variant={variant} shard k {k_row} of {k}: encode (all {W * batch}) {t_enc:.2f} coarse {t_coarse:.2f} '
      f'shard scan ({W * batch} queries) {t_scan:.2f}' + (f' (with the full k: {t_scan_k:.2f})' if t_scan_k else '') +
      f' exchange compute {t_merge:.2f} (split {t_split:.2f} + heads {t_m1:.2f} + '
      f'held-back {t_x:.2f} + final {t_m2:.2f}; head width {kp}, queries asking {asked:.3f}, held-back keys per query {held:.1f}) '
      f'third phase {t_3:.2f}' + (f' (rescan of {n3} (query, shard) rows {t_rescan:.2f} + merge {t_m3:.2f} + rescoring of {q3} queries again {t_r3:.2f}; '
                                   f'third_phase_queries {q3 / batch:.5f} of the batch)' if third and nkeep < k_row else '') +
      f'; equals the merge of the full rows: {exact}, overflow '
      f'{int(flag[0].item()) if flag is not None else 0}; the full-row merge it replaces {t_old:.2f}; rescore {t_resc:.2f} '
      f'| compute per step {tot:.2f} ms (collectives excluded)')

# how the final top-k spreads over the shards: per query the largest number of its k best hits
# that ONE shard holds (what a shard-side k smaller than k has to cover)
rows = torch.stack(rows_full)                                   # [W, batch, k] packed keys
flip = torch.tensor(-2 ** 63, dtype=torch.int64, device=dev)   # unsigned order as signed order
allk = (rows ^ flip).permute(1, 0, 2).reshape(batch, -1)
kth = torch.topk(allk, k, dim=1).values[:, -1:]                 # the k-th best key of the union
share = ((rows ^ flip) >= kth.unsqueeze(0)).sum(2)              # [W, batch]
mx = share.max(0).values.float()
qs = torch.quantile(mx, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev)).tolist()
print(f'largest share of a query\'s top {k} in one shard: median {qs[0]:.0f}, 90 % {qs[1]:.0f}, 99 % {qs[2]:.0f}, '
      f'99.9 % {qs[3]:.0f}, max {qs[4]:.0f}; queries with a shard holding more than 512: {(mx > 512).float().mean():.5f}, '
      f'more than 384: {(mx > 384).float().mean():.5f}, more than 640: {(mx > 640).float().mean():.5f}')

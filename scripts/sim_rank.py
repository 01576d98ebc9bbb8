"""Measurement helper: the compute one rank of a W-GPU sharded search performs per step,
emulated on ONE GPU (shard 0 of W, W x batch queries). Collectives are not included.

  python scripts/sim_rank.py W [library_size] [batch] [scan_variant] [ivfpq|ivfflat] [shard_k (default asl_shard_k(k, W); k = the round-4 protocol)] [dense]
(`dense`: the other ranks' queries as dense rows, as before the entry lists)
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
cfg = Config.open_search(num_list=4096, num_probe=128, num_candidates=1024, index=index, pq_m=32,
             kmeans_niter=25, mode='ann', precursor_tolerance_mass_open=500.0,
             precursor_tolerance_mode_open='Da', batch_size=batch, seed=1234)
sl = SpectralLibrary(lib, config=cfg, device=dev)
idx = sl._get_ann_index(2)
idx.set_scan_variant(variant)
q_all, _ = synthetic.make_queries(lib, aux, W * batch, seed=42, open_range=500.0, charge=2)
q = q_all.select(torch.arange(batch, device=dev)).contiguous()
be = HipShardBackend(sl, 2, 'open')
from ann_solo_amd import _lib
from ann_solo_amd.distributed import head_width, shard_k
k = be.k_scan
# the shards' own k: asl_shard_k by default (k / 2 at 8 ranks, 5 k / 8 at 4; second scans on the shard
# keep the result exact); argv[6] overrides it (k itself: the round-4 protocol)
k_row = int(sys.argv[6]) if len(sys.argv) > 6 and int(sys.argv[6]) > 0 else shard_k(k, W)
entries = not (len(sys.argv) > 7 and sys.argv[7] == 'dense')
kp = head_width(k, W)
if not (kp - 1 < k_row < k):
    k_row = k
shard_side = k_row < k
two_phase = kp - 1 < k_row
allvec = be.encode(q_all)
cD, cI = be.coarse(allvec)
nall = W * batch
# the W shards of the index, side by side on this GPU (copies through a file). Every shard scans ALL
# W x batch queries (that is its work per step); kept: the heads of all its rows (what the owners
# receive), and for this rank's own `batch` queries its rows, floors, smallest keys and FULL rows
import tempfile
from ann_solo_amd import faiss_compat as faiss
tmp = os.path.join(tempfile.mkdtemp(), 'sim.idxmi')
faiss.write_index(idx, tmp)
heads_all, own = [None] * W, [None] * W
for s_ in range(1, W):
    other = faiss.read_index(tmp)
    other.shard(s_, W)
    rows_s = other.search_preassigned_keys(allvec, k_row, cD, cI)
    if two_phase:
        h_, f_, m_ = be.keys_split(rows_s, kp, True)
        heads_all[s_] = h_
        own[s_] = (rows_s[:batch].contiguous(), f_[:batch].contiguous(), m_[:batch].contiguous(),
                   other.search_preassigned_keys(allvec[:batch], k, cD[:batch], cI[:batch]) if shard_side else None)
    else:
        own[s_] = (rows_s[:batch].contiguous(), None, None, None)
    del other, rows_s
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


# every rank hashes ALL queries (peaks travel): into entry lists -- what the scans read -- or dense rows
vec = be.encode(q)
if entries:
    def enc_all():
        be.encode(q)                            # the own slice as dense rows (the coarse quantiser's input)
        return be.encode_entries(q_all)
    t_enc, allq = timed(enc_all)
else:
    t_enc, allq = timed(lambda: be.encode(q_all))
t_coarse, _ = timed(lambda: be.coarse(vec))     # ... and quantises its own slice
t_scan, K = timed(lambda: be.shard_search_keys(allq, cD, cI, k=k_row))
t_scan_k = None
if shard_side:
    t_scan_k, _ = timed(lambda: be.shard_search_keys(allq, cD, cI, k=k))
full0 = be.shard_search_keys(allvec[:batch], cD[:batch], cI[:batch], k=k)
t_split = t_m1 = t_x = t_m2 = 0.0
rescans, asked, held, overflow, exact = 0, 0.0, 0.0, 0, True
if two_phase:
    t_split, (head, floor, rowmin) = timed(lambda: be.keys_split(K, kp, True) if shard_side else be.keys_split(K, kp) + (None,))
    heads_all[0] = head
    own[0] = (K[:batch].contiguous(), floor[:batch].contiguous(), None if rowmin is None else rowmin[:batch].contiguous(), full0)
    # owner side: this rank merges the heads of its own `batch` queries (timed); the bounds of ALL
    # queries (what the W owners send to shard 0) come from the same merge over everything
    heads_own = torch.stack([h_[:batch] for h_ in heads_all]).contiguous()
    t_m1, (okeys, bnd, need) = timed(lambda: be.keys_merge_heads(heads_own, k))
    bnd_all = torch.cat([be.keys_merge_heads(torch.stack([h_[o_ * batch:(o_ + 1) * batch] for h_ in heads_all]).contiguous(), k)[1][0]
                         for o_ in range(W)])                     # [W * batch]: the bounds shard 0 is sent
    del heads_all
    xcap = batch * max(8, k // 16)
    # shard side, exactly this rank's work: the answers to all W owners, with the second scans
    def shard_side_step():
        f_ = be.new_flag()
        x_ = be.keys_extras(K, floor, bnd_all, W, xcap, f_, rescan=(rowmin, allq, cD, cI, k) if shard_side else None)
        return x_, f_
    t_x, (xbuf0, f0) = timed(shard_side_step)
    rescans, overflow = int(f0[1].item()), int(f0[0].item())
    # owner side: the real answers of the W shards for the own queries (shard 0's from the step
    # above; the other shards' second scans answered from their full rows)
    xbufs = [xbuf0[0].contiguous()]
    flag = be.new_flag()
    L = _lib.lib()
    for s_ in range(1, W):
        rows_s, f_s, m_s, full_s = own[s_]
        b_ = bnd[s_].contiguous()
        if shard_side:
            R = batch
            rowlist = torch.zeros(R, dtype=torch.int64, device=dev)
            rmap = torch.empty(batch, dtype=torch.int32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(L.asl_keys_rescan_list(batch, _lib.ptr(b_), _lib.ptr(m_s), R, _lib.ptr(rowlist), _lib.ptr(rmap),
                                              _lib.ptr(cnt), _lib.ptr(flag)))
            K3 = full_s.index_select(0, rowlist)
            x_ = torch.empty((1, batch + xcap), dtype=torch.int64, device=dev)
            cur = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(L.asl_keys_extras(1, batch, k_row, _lib.ptr(rows_s), _lib.ptr(f_s), _lib.ptr(b_), xcap, _lib.ptr(x_),
                                         _lib.ptr(cur), _lib.ptr(flag), _lib.ptr(rmap), _lib.ptr(K3), k))
            xbufs.append(x_[0])
        else:
            xbufs.append(be.keys_extras(rows_s, f_s, b_, 1, xcap, flag)[0])
    xbuf = torch.stack(xbufs)
    overflow |= int(flag[0].item())
    t_m2, knn = timed(lambda: be.keys_merge_final(heads_own, xbuf, okeys, need, k))
    want = be.merge_keys(torch.stack([o_[3] if shard_side else o_[0] for o_ in own]).contiguous())[1]
    exact = bool(torch.equal(torch.sort(knn, 1).values, torch.sort(want, 1).values))
    held = float(((xbuf[:, :batch] >> 32).sum()).item()) / batch
    asked = float(need.float().mean())
else:
    own[0] = (K[:batch].contiguous(), None, None, None)
    t_m2, (_, knn) = timed(lambda: be.merge_keys(torch.stack([o_[0] for o_ in own]).contiguous()))
t_old, _ = timed(lambda: be.merge_keys(K.view(W, batch, -1).contiguous()))
t_resc, _ = timed(lambda: be.rescore_knn(q, knn, True))
t_merge = t_split + t_m1 + t_x + t_m2
tot = t_enc + t_coarse + t_scan + t_merge + t_resc
print(f'{index} W={W} batch/rank={batch} variant={variant} shard k {k_row} of {k}, queries as {"entry lists" if entries else "dense rows"}: encode (all {nall}) {t_enc:.2f} coarse {t_coarse:.2f} '
      f'shard scan ({nall} queries) {t_scan:.2f}' + (f' (with the full k: {t_scan_k:.2f})' if t_scan_k else '') +
      f' exchange compute {t_merge:.2f} (split {t_split:.2f} + heads {t_m1:.2f} + '
      f'answers incl. second scans {t_x:.2f} + final {t_m2:.2f}; head width {kp}, queries asking {asked:.3f}, '
      f'answer keys per query {held:.1f}, rows this shard scanned a second time {rescans} = {rescans / nall:.5f} of its rows); '
      f'equals the merge of the full rows: {exact}, overflow {overflow}; the full-row merge it replaces {t_old:.2f}; '
      f'rescore {t_resc:.2f} | compute per step {tot:.2f} ms (collectives excluded)')

# how the final top-k spreads over the shards: per query the largest number of its k best hits
# that ONE shard holds (what a shard-side k smaller than k has to cover)
if shard_side:
    rows = torch.stack([o_[3] for o_ in own])                       # [W, batch, k] packed keys
    flip = torch.tensor(-2 ** 63, dtype=torch.int64, device=dev)   # unsigned order as signed order
    allk = (rows ^ flip).permute(1, 0, 2).reshape(batch, -1)
    kth = torch.topk(allk, k, dim=1).values[:, -1:]                 # the k-th best key of the union
    share = ((rows ^ flip) >= kth.unsqueeze(0)).sum(2)              # [W, batch]
    mx = share.max(0).values.float()
    qs = torch.quantile(mx, torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], device=dev)).tolist()
    print(f'largest share of a query\'s top {k} in one shard: median {qs[0]:.0f}, 90 % {qs[1]:.0f}, 99 % {qs[2]:.0f}, '
          f'99.9 % {qs[3]:.0f}, max {qs[4]:.0f}; queries with a shard holding more than {k_row}: {(mx > k_row).float().mean():.5f}')

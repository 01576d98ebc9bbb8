"""Randomised check of list sharding on ONE GPU: an index is saved, loaded W times, each copy
sharded (asl_index_shard) and searched with the caller's probe lists -- exact top-k sets, packed
keys when the scan supports them, (D, I) rows otherwise -- and the merge of the W partial results
must equal the unsharded search bit for bit (ids and scores), for random index kinds, PQ shapes,
W, nlist, nprobe and k.   python scripts/fuzz_shards.py [seconds] [seed]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from ann_solo_amd import _lib, faiss_compat as faiss, synthetic
from ann_solo_amd.distributed import HipShardBackend, head_width
from ann_solo_amd.spectral_library import Config, SpectralLibrary

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
t_end = time.time() + budget
trials = bad = two_phase = third_runs = third_asked = entry_runs = 0
dev = torch.device('cuda', 0)
tmp = tempfile.mkdtemp()
while time.time() < t_end:
    n = int(rng.choice([3000, 12000, 50000]))
    nlist = int(rng.choice([8, 32, 128, 128, 1100]))      # 1100: up to 1 100 probes (> 512: two probes per thread; > 1024: generic)
    nlist = min(nlist, n // 40)
    nprobe = int(rng.integers(1, nlist + 1))
    k = int(rng.choice([1, 50, 256, 1024, 1280, 2048]))
    W = int(rng.choice([2, 3, 4, 8]))
    index = str(rng.choice(['ivfpq', 'ivfflat']))
    pq_m = int(rng.choice([16, 32]))
    nq = int(rng.choice([5, 200, 900]))
    desc = dict(n=n, nlist=nlist, nprobe=nprobe, k=k, W=W, index=index, pq_m=pq_m, nq=nq)
    lib, aux = synthetic.make_library(n, seed=int(rng.integers(1, 1 << 30)), device='cpu', charges=(2,),
                                      charge_p=(1.0,))
    q, _ = synthetic.make_queries(lib, aux, nq, seed=int(rng.integers(1, 1 << 30)), charge=2)
    sl = SpectralLibrary(lib, config=Config.open_search(num_list=nlist, num_probe=nprobe, num_candidates=k, index=index,
                                            pq_m=pq_m, kmeans_niter=2), device=dev)
    idx = sl._get_ann_index(2)
    idx.nprobe = nprobe
    vec = sl._encode(q.to(dev))
    D, I = idx.search(vec, k)
    cD, cI = idx.coarse(vec, nprobe)
    path = os.path.join(tmp, 'x.idxmi')
    faiss.write_index(idx, path)
    owner = idx.shard_map(W)
    parts, keys, keys_s, nloc = [], [], [], 0
    ok_e = True
    use_keys = bool(_lib.lib().asl_index_supports_keys(idx._h, k, nprobe))
    # a random head and a random shard-side k between the head and k (second scans answered from the full rows)
    hk = int(rng.integers(1, k + 1)) if rng.random() < 0.7 else min(k, -(-2 * k // W))
    ks = int(rng.integers(hk + 1, k)) if hk + 1 < k and rng.random() < 0.6 else k
    # half of the runs: the shards search the queries as ENTRY LISTS (asl_encode_entries_batch ->
    # asl_index_search_entries) instead of dense rows -- the same packed-key rows are expected
    be0 = HipShardBackend.__new__(HipShardBackend)
    be0.sl, be0.device = sl, dev
    eq = be0.encode_entries(q) if rng.random() < 0.5 else None
    entry_runs += eq is not None
    for r in range(W):
        sh = faiss.read_index(path)
        sh.shard(r, W)
        nloc += sh.info().nlocal
        if use_keys:
            use_keys = bool(_lib.lib().asl_index_supports_keys(sh._h, k, nprobe))      # (an empty / dense shard)
        if use_keys:
            if eq is not None:
                keys.append(sh.search_entries_keys(eq.entries, eq.counts, k, cD, cI))
                keys_s.append(sh.search_entries_keys(eq.entries, eq.counts, ks, cD, cI) if ks < k else keys[-1])
                if r == 0:      # ... and they ARE the dense call's rows (as sets)
                    ok_e = torch.equal(keys[-1].sort(1).values, sh.search_preassigned_keys(vec, k, cD, cI).sort(1).values)
            else:
                keys.append(sh.search_preassigned_keys(vec, k, cD, cI))
                keys_s.append(sh.search_preassigned_keys(vec, ks, cD, cI) if ks < k else keys[-1])
        sh.set_unordered(True)
        parts.append(sh.search_preassigned(vec, k, cD, cI))
        del sh
    Dm, Im = faiss.topk_merge(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    ok = nloc == n and torch.equal(Im, I) and torch.equal(Dm.view(torch.int32), D.view(torch.int32)) and ok_e
    ok &= set(owner.tolist()) <= set(range(W))
    if use_keys and len(keys) == W:
        Dk, Ik = faiss.topk_merge_keys(torch.stack(keys))
        ok &= torch.equal(Ik, I) and torch.equal(Dk.view(torch.int32), D.view(torch.int32))
        if k + 768 <= 2048:
            # the two-phase exchange (csrc/exchange.hip) over the same rows, a random head width: heads
            # -> bounds -> held-back keys -> final merge must give the unsharded ids as a set
            be = HipShardBackend.__new__(HipShardBackend)
            shard_side = hk < ks < k
            kp = head_width(k, W, hk)
            rows = keys_s if shard_side else keys
            split = [be.keys_split(K_, kp, True) for K_ in rows]
            heads = torch.stack([h for h, _, _ in split])
            okeys, bnd, need = be.keys_merge_heads(heads, k)
            If = None
            if hk < k:
                flag = torch.zeros(2, dtype=torch.int32, device=dev)
                xcap = nq * int(rng.choice([k, max(8, k // 16)]))
                xb = []
                for r in range(W):
                    b_ = bnd[r].contiguous()
                    if shard_side:
                        # rows of ks < k keys: where the bound lies below the smallest key of a full row
                        # the shard answers from its FULL row (what a second scan with k returns)
                        R = max(64, nq // 16) if rng.random() < 0.8 else nq
                        rowlist = torch.zeros(R, dtype=torch.int64, device=dev)
                        rmap = torch.empty(nq, dtype=torch.int32, device=dev)
                        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
                        _lib.check(_lib.lib().asl_keys_rescan_list(nq, _lib.ptr(b_), _lib.ptr(split[r][2]), R,
                                                                   _lib.ptr(rowlist), _lib.ptr(rmap), _lib.ptr(cnt),
                                                                   _lib.ptr(flag)))
                        flag[1:2] += cnt
                        K3 = keys[r].index_select(0, rowlist)
                        x_ = torch.empty((1, nq + xcap), dtype=torch.int64, device=dev)
                        cur = torch.zeros(1, dtype=torch.int32, device=dev)
                        _lib.check(_lib.lib().asl_keys_extras(1, nq, ks, _lib.ptr(rows[r]), _lib.ptr(split[r][1]),
                                                              _lib.ptr(b_), xcap, _lib.ptr(x_), _lib.ptr(cur),
                                                              _lib.ptr(flag), _lib.ptr(rmap), _lib.ptr(K3), k))
                        xb.append(x_[0])
                    else:
                        xb.append(be.keys_extras(rows[r], split[r][1], b_, 1, xcap, flag)[0])
                if not int(flag[0].item()):
                    If = be.keys_merge_final(heads, torch.stack(xb), okeys, need, k)
                    if shard_side:
                        third_runs += 1
                        third_asked += int(flag[1].item())
            else:
                If = be.keys_merge_final(heads, None, okeys, need, k)
            if If is not None:
                ok &= torch.equal(torch.sort(If, 1).values, torch.sort(I, 1).values)
                two_phase += 1
    trials += 1
    if not ok:
        bad += 1
        print('MISMATCH', desc, flush=True)
    sl.shutdown()
print(f'{trials} trials ({two_phase} also through the two-phase exchange, {third_runs} of them with a shard-side '
      f'k < k: {third_asked} rows answered from a second scan; {entry_runs} with the queries as entry lists), '
      f'{bad} mismatches')

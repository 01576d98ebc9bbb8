"""The device kernels of the exact top-k exchange (ann_solo_amd/csrc/exchange.hip) against their
numpy restatement (tests/exchange_ref.py) and against the definition: the ids the owner ends up
with are the top-k of the union of all shards' rows, whatever the head width -- and, with second
scans on the shards, whatever the shards' own k."""
import numpy as np
import pytest

import exchange_ref as X

pytestmark = pytest.mark.gpu


def _rows(rng, nrows, k, fill=1.0, lo=-0.1, hi=0.95, grid=None, id_base=0, id_span=1 << 20):
    """[nrows, k] packed keys: random scores (optionally on a coarse grid: many ties and crowded
    buckets), unique ids per row, a ragged number of valid entries, shuffled (rows are sets)."""
    K = np.zeros((nrows, k), np.uint64)
    for r in range(nrows):
        n = k if fill >= 1.0 else int(rng.integers(0, k + 1) if rng.random() < 0.5 else k)
        s = rng.uniform(lo, hi, n).astype(np.float32)
        if grid:
            s = (np.round(s * grid) / grid).astype(np.float32)
        ids = id_base + rng.choice(id_span, n, replace=False)
        keys = X.pack_keys(s, ids).view(np.uint64)
        rng.shuffle(keys)
        K[r, :n] = keys
    return K.view(np.int64)


def _sets(a):
    a = np.asarray(a).view(np.uint64)
    return [set(r[r != 0].tolist()) for r in a]


@pytest.mark.parametrize('k,kp', [(1024, 257), (1024, 513), (64, 17), (100, 101), (1280, 2), (256, 33)])
def test_keys_split_equals_the_restatement(k, kp):
    import torch
    from ann_solo_amd.distributed import HipShardBackend
    be = HipShardBackend.__new__(HipShardBackend)
    rng = np.random.default_rng(k + kp)
    K = np.concatenate([_rows(rng, 40, k, fill=0.5), _rows(rng, 24, k, grid=50),
                        np.zeros((2, k), np.int64)])
    head, floors, rowmin = be.keys_split(torch.from_numpy(K).cuda(), kp, True)
    h1, f1 = be.keys_split(torch.from_numpy(K).cuda(), kp)
    assert torch.equal(h1.sort(1).values, head.sort(1).values) and torch.equal(f1, floors)
    head, floors = head.cpu().numpy(), floors.cpu().numpy()
    h0, f0 = X.keys_split(K, kp)
    Ku = K.view(np.uint64)                                                    # M: the smallest key of a FULL row
    full = (Ku != 0).all(1)
    assert full.any() and (~full).any()
    assert np.array_equal(rowmin.cpu().numpy().view(np.uint64), np.where(full, Ku.min(1), 0))
    assert np.array_equal(rowmin.cpu().numpy(), X.row_min(K))
    assert np.array_equal(head[:, kp - 1], h0[:, kp - 1])                     # best held-back key
    assert np.array_equal(floors, f0)                                         # the bucket floor of every row
    assert _sets(head[:, :kp - 1]) == _sets(h0[:, :kp - 1])
    rest = X.held_back(K, floors)                                             # what stays behind in K
    for r in range(len(K)):                                                   # zero padding is at the end
        nz = np.nonzero(head[r, :kp - 1])[0]
        assert len(nz) == 0 or nz[-1] == len(nz) - 1
        a, h = _sets(head[r:r + 1, :kp - 1])[0], _sets(rest[r:r + 1])[0]
        assert not h or not a or max(h) < min(a)                              # kept keys beat held-back ones
        assert len(a) <= kp - 1 and a | h == _sets(K[r:r + 1])[0]


@pytest.mark.parametrize('S,k,head_keys,xper,grid', [(8, 1024, 256, 64, None), (4, 256, 40, 256, None),
                                                     (3, 100, 100, 0, None), (8, 512, 16, 512, 40),
                                                     (2, 64, 1, 64, None), (5, 1280, 300, 8, None),
                                                     (3, 1024, 800, 256, None), (16, 1024, 128, 128, 60),
                                                     # threshold buckets of ~800 / ~340 / ~160 keys: the finish's
                                                     # serial ranking (more keys than threads) and its parallel
                                                     # one at several group sizes
                                                     (8, 512, 64, 512, 5), (8, 1024, 256, 1024, 12),
                                                     (4, 1024, 512, 1024, 25)])
def test_two_phase_exchange_is_the_top_k_of_the_union(S, k, head_keys, xper, grid):
    """One owner, S shards, all on one device: split -> merge of the heads -> bounds -> held-back
    keys -> final merge. Every step equals the restatement; the result equals the definition.
    (S x head keys <= 2048: the merge's bulk load; 3 x 800: its streaming offers.)"""
    import torch
    from ann_solo_amd.distributed import HipShardBackend, head_width
    be = HipShardBackend.__new__(HipShardBackend)
    rng = np.random.default_rng(S * k + head_keys)
    n = 37
    # shard s owns ids [s << 20, (s + 1) << 20): keys are unique across shards; skewed scores so that
    # some shards hold far more than their share of a query's best hits
    rows = [_rows(rng, n, k, fill=0.7, hi=0.5 + 0.45 * rng.random(), grid=grid, id_base=s << 20) for s in range(S)]
    kp = head_width(k, S, head_keys)
    heads, floors = [], []
    for s in range(S):
        h, r = be.keys_split(torch.from_numpy(rows[s]).cuda(), kp)
        heads.append(h)
        floors.append(r)
    heads = torch.stack(heads)                                   # what the owner receives [S, n, kp]
    out, bounds, need = be.keys_merge_heads(heads, k)
    o0, b0, n0 = X.keys_merge_heads(heads.cpu().numpy(), k)
    assert _sets(out.cpu().numpy()) == _sets(o0)
    assert np.array_equal(bounds.cpu().numpy(), b0) and np.array_equal(need.cpu().numpy(), n0)
    union = np.concatenate(rows, axis=1).view(np.uint64)
    want = [set(X.key_id(np.sort(u[u != 0])[::-1][:k]).tolist()) for u in union]
    if kp - 1 >= k:                                              # nothing can be held back
        assert not need.any()
        I = be.keys_merge_final(heads, None, out, need, k)
    else:
        # every shard answers the questions addressed to it: "destination" = the one owner, so
        # the shard-side call runs with world = 1 rows-per-destination = n
        xcap = n * xper
        flag = torch.zeros(2, dtype=torch.int32, device='cuda')
        dev_rows = [torch.from_numpy(rows[s]).cuda() for s in range(S)]
        xbufs = [be.keys_extras(dev_rows[s], floors[s], bounds[s].contiguous(), 1, xcap, flag)[0] for s in range(S)]
        overflow = int(flag[0].item())
        ref_over = 0
        for s in range(S):
            xb0, ov = X.keys_extras(rows[s], floors[s].cpu().numpy(), b0[s], 1, xcap)
            ref_over |= ov
            got, exp = xbufs[s].cpu().numpy().view(np.uint64), xb0[0].view(np.uint64)
            if not ov:
                for q in range(n):                               # same keys per query, wherever they sit
                    c, st = int(got[q] >> np.uint64(32)), int(got[q] & np.uint64(0xFFFFFFFF))
                    c0, st0 = int(exp[q] >> np.uint64(32)), int(exp[q] & np.uint64(0xFFFFFFFF))
                    assert c == c0 and set(got[n + st:n + st + c].tolist()) == set(exp[n + st0:n + st0 + c0].tolist())
        assert overflow == ref_over
        if overflow:
            assert xper < k            # only a buffer smaller than a row can run full here
            return
        I = be.keys_merge_final(heads, torch.stack(xbufs), out, need, k)
    I = I.cpu().numpy()
    for q in range(n):
        got = I[q][I[q] >= 0]
        assert len(got) == len(set(got.tolist())) == len(want[q]) and set(got.tolist()) == want[q], q
        assert (I[q][len(got):] == -1).all()


@pytest.mark.parametrize('index', ['ivfpq', 'ivfflat'])
def test_gated_search_scans_exactly_the_counted_rows(index):
    """asl_index_search_gated: a launch for `cap` rows of which a DEVICE-side count says how many are
    real -- those rows equal search_preassigned's packed-key rows, the others stay untouched; and
    HipShardBackend.keys_extras(rescan=...) through a real (sharded) index equals the restatement."""
    import torch
    from ann_solo_amd import _lib, synthetic
    from ann_solo_amd.distributed import HipShardBackend
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(20000, seed=5, device='cpu', charges=(2,), charge_p=(1.0,))
    q, _ = synthetic.make_queries(lib, aux, 300, seed=6, charge=2)
    sl = SpectralLibrary(lib, config=Config.open_search(num_list=32, num_probe=12, num_candidates=256, index=index, kmeans_niter=3))
    be = HipShardBackend(sl, 2, 'open')
    idx = be.index
    vec = be.encode(q)
    cD, cI = be.coarse(vec)
    k = 256
    want = idx.search_preassigned_keys(vec, k, cD, cI)
    for count in (0, 1, 77, 300):
        K = torch.full((300, k), -7, dtype=torch.int64, device=vec.device)
        cnt = torch.tensor([count], dtype=torch.int32, device=vec.device)
        idx.set_unordered(2)
        try:
            _lib.check(_lib.lib().asl_index_search_gated(idx._h, 300, _lib.ptr(vec), k, 12, _lib.ptr(cD), _lib.ptr(cI),
                                                         None, _lib.ptr(K), _lib.ptr(cnt)))
        finally:
            idx.set_unordered(0)
        assert torch.equal(K[:count].sort(1).values, want[:count].sort(1).values)
        assert bool((K[count:] == -7).all())
    # the whole shard-side step on a real shard: rows of 64 keys, bounds that ask about a third of them
    idx.shard(0, 2)
    ks = 64
    Ks = idx.search_preassigned_keys(vec, ks, cD, cI)
    Kf = idx.search_preassigned_keys(vec, k, cD, cI)
    head, floor, rowmin = be.keys_split(Ks, 17, True)
    Kn = Ks.cpu().numpy().view(np.uint64)
    rng = np.random.default_rng(3)
    b = np.full(300, X.NONE, np.uint64)
    for r in range(300):
        row = np.sort(Kn[r][Kn[r] != 0])
        u = rng.random()
        if len(row) and u < 0.3:
            b[r] = row[0] - np.uint64(1 + rng.integers(0, 1 << 30))      # below the smallest key: second scan if full
        elif len(row) and u < 0.6:
            b[r] = row[len(row) // 2]                                        # inside the row: held-back keys only
    bounds = torch.from_numpy(b.view(np.int64)).cuda()
    flag = be.new_flag()
    xcap = 300 * k
    be.rescan_capacity = 300
    xbuf = be.keys_extras(Ks, floor, bounds, 1, xcap, flag, rescan=(rowmin, vec, cD, cI, k))
    assert int(flag[0].item()) == 0 and int(flag[1].item()) > 10
    rl0, rm0, c0, _ = X.rescan_list(b, rowmin.cpu().numpy(), 300)
    assert int(flag[1].item()) == c0
    K3 = np.zeros((300, k), np.int64)
    K3[:c0] = Kf.cpu().numpy()[rl0[:c0]]
    x0, ov = X.keys_extras(Ks.cpu().numpy(), floor.cpu().numpy(), b, 1, xcap, rm0, K3)
    got, exp = xbuf[0].cpu().numpy().view(np.uint64), x0[0].view(np.uint64)
    for r in range(300):
        c, st = int(got[r] >> np.uint64(32)), int(got[r] & np.uint64(0xFFFFFFFF))
        c1, st1 = int(exp[r] >> np.uint64(32)), int(exp[r] & np.uint64(0xFFFFFFFF))
        assert c == c1 and set(got[300 + st:300 + st + c].tolist()) == set(exp[300 + st1:300 + st1 + c1].tolist()), r
    sl.shutdown()


@pytest.mark.parametrize('index,storage', [('ivfpq', None), ('ivfflat', 'fp32'), ('ivfflat', 'fx22')])
def test_entry_list_search_equals_the_dense_search(index, storage):
    """asl_index_search_entries (queries as the entry lists of asl_encode_entries_batch) returns the
    packed-key rows of asl_index_search_preassigned on the dense rows, bit for bit -- whole batch,
    gated by a device-side count, on a sharded index -- and HipShardBackend.keys_extras answers the
    same from second scans of entry-list queries as from dense ones."""
    import torch
    from ann_solo_amd import synthetic
    from ann_solo_amd.distributed import EntryQueries, HipShardBackend
    from ann_solo_amd.spectral_library import Config, SpectralLibrary
    lib, aux = synthetic.make_library(20000, seed=15, device='cpu', charges=(2,), charge_p=(1.0,))
    q, _ = synthetic.make_queries(lib, aux, 300, seed=16, charge=2)
    kw = dict(flat_storage=storage) if storage else {}
    sl = SpectralLibrary(lib, config=Config.open_search(num_list=32, num_probe=12, num_candidates=256, index=index,
                                                        kmeans_niter=3, **kw))
    be = HipShardBackend(sl, 2, 'open')
    idx = be.index
    vec = be.encode(q)
    eq = be.encode_entries(q)
    assert isinstance(eq, EntryQueries) and eq.shape == (300,) and int(eq.counts.min()) > 0
    # the entry lists ARE the rows' non-zeros
    v = vec.cpu().numpy()
    e, c = eq.entries.cpu().numpy(), eq.counts.cpu().numpy()
    for r in (0, 17, 299):
        nz = np.nonzero(v[r])[0]
        assert c[r] == len(nz) and np.array_equal(e[r, :len(nz), 0], nz * 128)
        assert np.array_equal(e[r, :len(nz), 1], v[r, nz].view(np.int32))
    cD, cI = be.coarse(vec)
    for shard in (None, (1, 3)):
        if shard:
            idx.shard(*shard)
        for k in (256, 64):
            want = idx.search_preassigned_keys(vec, k, cD, cI)
            got = be.shard_search_keys(eq, cD, cI, k=k)
            assert torch.equal(got.sort(1).values, want.sort(1).values)
        sub = torch.arange(299, -1, -3, device=vec.device)            # a selection of rows, as a piece takes them
        got = be.shard_search_keys(eq.index_select(0, sub), cD[sub].contiguous(), cI[sub].contiguous(), k=256)
        assert torch.equal(got.sort(1).values, idx.search_preassigned_keys(vec, 256, cD, cI)[sub].sort(1).values)
        want = idx.search_preassigned_keys(vec, 256, cD, cI)
        for count in (0, 5, 300):
            cnt = torch.tensor([count], dtype=torch.int32, device=vec.device)
            K = idx.search_entries_keys(eq.entries, eq.counts, 256, cD, cI, gate=cnt)
            assert torch.equal(K[:count].sort(1).values, want[:count].sort(1).values)
    # second scans inside keys_extras: entry-list inputs give the dense inputs' answer buffer
    ks, k = 64, 256
    Ks = idx.search_preassigned_keys(vec, ks, cD, cI)
    head, floor, rowmin = be.keys_split(Ks, 17, True)
    Kn = Ks.cpu().numpy().view(np.uint64)
    rng = np.random.default_rng(4)
    b = np.full(300, X.NONE, np.uint64)
    for r in range(300):
        row = np.sort(Kn[r][Kn[r] != 0])
        if len(row) and rng.random() < 0.4:
            b[r] = row[0] - np.uint64(1 + rng.integers(0, 1 << 30))
    bounds = torch.from_numpy(b.view(np.int64)).cuda()
    be.rescan_capacity = 300
    out = []
    for xv in (vec, eq):
        flag = be.new_flag()
        xbuf = be.keys_extras(Ks, floor, bounds, 1, 300 * k, flag, rescan=(rowmin, xv, cD, cI, k))
        assert int(flag[0].item()) == 0
        out.append((xbuf[0].cpu().numpy().view(np.uint64), int(flag[1].item())))
    (a, na), (bb, nb) = out
    assert na == nb and na > 10
    for r in range(300):
        ca, sa = int(a[r] >> np.uint64(32)), int(a[r] & np.uint64(0xFFFFFFFF))
        cb, sb = int(bb[r] >> np.uint64(32)), int(bb[r] & np.uint64(0xFFFFFFFF))
        assert ca == cb and set(a[300 + sa:300 + sa + ca].tolist()) == set(bb[300 + sb:300 + sb + cb].tolist()), r
    # a row with more than 64 non-zeros and no dense form is searched as an all-zero query, never read
    cnt_bad = eq.counts.clone()
    cnt_bad[3] = -70
    K = idx.search_entries_keys(eq.entries, cnt_bad, 64, cD, cI)
    torch.cuda.synchronize()
    assert torch.equal(K[4:].sort(1).values, idx.search_preassigned_keys(vec, 64, cD, cI)[4:].sort(1).values)
    sl.shutdown()


@pytest.mark.parametrize('S,k,ks,head_keys,xper,grid,skew,cap', [
    (8, 1024, 512, 256, 1024, None, 0.45, None), (8, 1024, 512, 256, 1024, None, 0.9, None),
    (4, 1024, 640, 512, 1024, None, 0.6, None),
    (8, 1024, 128, 64, 1024, None, 0.2, None),        # 8 x 128 = k: nearly every full row is scanned again
    (3, 256, 64, 16, 256, 30, 0.5, None), (8, 512, 192, 128, 512, 8, 0.5, None), (2, 64, 16, 8, 64, None, 0.3, None),
    (8, 1024, 128, 64, 1024, None, 0.2, 3)])          # room for 3 second scans: overflow
def test_shard_side_k_with_second_scans_is_the_top_k_of_the_union_of_the_full_rows(S, k, ks, head_keys, xper, grid, skew, cap):
    """The shards hold FULL rows of k keys but run the exchange on their k_s best (what a scan with
    a shard-side k_s keeps). Heads -> bounds -> rescan list (bound below the smallest key of a full
    row) -> answers from the full row for those, from the k_s-row for the others -> final merge:
    every step equals the restatement, the result is the top k of the union of the FULL rows."""
    import torch
    from ann_solo_amd import _lib
    from ann_solo_amd.distributed import HipShardBackend
    be = HipShardBackend.__new__(HipShardBackend)
    rng = np.random.default_rng(S * k + ks)
    n = 41
    full = [_rows(rng, n, k, fill=0.8, hi=0.5 + skew * rng.random(), grid=grid, id_base=s << 20) for s in range(S)]

    def top(K, m):          # the m best keys of every row, shuffled (rows are sets), 0 padded
        Ku = np.sort(K.view(np.uint64), 1)[:, ::-1][:, :m].copy()
        for r in Ku:
            rng.shuffle(r)
        return Ku.view(np.int64)
    rows = [top(f, ks) for f in full]
    kp = head_keys + 1
    dev_rows = [torch.from_numpy(r).cuda() for r in rows]
    split = [be.keys_split(r, kp, True) for r in dev_rows]
    heads = torch.stack([h for h, _, _ in split])
    out, bounds, need = be.keys_merge_heads(heads, k)
    o0, b0, n0 = X.keys_merge_heads(heads.cpu().numpy(), k)
    assert np.array_equal(bounds.cpu().numpy(), b0) and np.array_equal(need.cpu().numpy(), n0)
    flag = torch.zeros(2, dtype=torch.int32, device='cuda')
    xcap = n * xper
    R = cap or max(64, n // 16)
    L = _lib.lib()
    xbufs, rescans, over_ref = [], 0, 0
    for s in range(S):
        # the device steps of keys_extras(rescan=...) one by one, the second scan answered from the full rows
        bnd = bounds[s].contiguous()
        rowlist = torch.zeros(R, dtype=torch.int64, device='cuda')
        rmap = torch.empty(n, dtype=torch.int32, device='cuda')
        cnt = torch.zeros(1, dtype=torch.int32, device='cuda')
        _lib.check(L.asl_keys_rescan_list(n, _lib.ptr(bnd), _lib.ptr(split[s][2]), R, _lib.ptr(rowlist), _lib.ptr(rmap),
                                          _lib.ptr(cnt), _lib.ptr(flag)))
        rl0, rm0, c0, ov0 = X.rescan_list(b0[s], split[s][2].cpu().numpy(), R)
        over_ref |= ov0
        assert int(cnt.item()) == c0
        got_rows = set(rowlist[:min(c0, R)].cpu().tolist())
        assert got_rows <= set(np.nonzero((b0[s].view(np.uint64) != X.NONE))[0].tolist()) and len(got_rows) == min(c0, R)
        if not ov0:
            assert got_rows == set(rl0[:c0].tolist())
            assert np.array_equal(rmap.cpu().numpy() >= 0, rm0 >= 0)
        rescans += min(c0, R)
        K3 = torch.from_numpy(full[s]).cuda().index_select(0, rowlist)
        xbuf = torch.empty((1, n + xcap), dtype=torch.int64, device='cuda')
        cursor = torch.zeros(1, dtype=torch.int32, device='cuda')
        _lib.check(L.asl_keys_extras(1, n, ks, _lib.ptr(dev_rows[s]), _lib.ptr(split[s][1]), _lib.ptr(bnd), xcap,
                                     _lib.ptr(xbuf), _lib.ptr(cursor), _lib.ptr(flag), _lib.ptr(rmap), _lib.ptr(K3), k))
        xbufs.append(xbuf[0])
        if not ov0:
            rm_dev = rmap.cpu().numpy()
            x0, ov = X.keys_extras(rows[s], split[s][1].cpu().numpy(), b0[s], 1, xcap, rm_dev,
                                   K3.cpu().numpy())
            over_ref |= ov
            if not ov:
                got, exp = xbuf[0].cpu().numpy().view(np.uint64), x0[0].view(np.uint64)
                for q in range(n):
                    c, st = int(got[q] >> np.uint64(32)), int(got[q] & np.uint64(0xFFFFFFFF))
                    c1, st1 = int(exp[q] >> np.uint64(32)), int(exp[q] & np.uint64(0xFFFFFFFF))
                    assert c == c1 and set(got[n + st:n + st + c].tolist()) == set(exp[n + st1:n + st1 + c1].tolist())
    assert int(flag[0].item()) == over_ref
    if over_ref:
        assert cap is not None or xper < k
        return
    assert rescans > 0 or ks * S > k
    I = be.keys_merge_final(heads, torch.stack(xbufs), out, need, k).cpu().numpy()
    union = np.concatenate(full, axis=1).view(np.uint64)
    want = [set(X.key_id(np.sort(u[u != 0])[::-1][:k]).tolist()) for u in union]
    for q in range(n):
        got = I[q][I[q] >= 0]
        assert len(got) == len(set(got.tolist())) == len(want[q]) and set(got.tolist()) == want[q], q
        assert (I[q][len(got):] == -1).all()

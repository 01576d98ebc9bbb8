"""The device kernels of the exact top-k exchange (ann_solo_amd/csrc/exchange.hip) against their
numpy restatement (tests/exchange_ref.py) and against the definition: the ids the owner ends up
with are the top-k of the union of all shards' rows, whatever the head width -- and, with the
third phase, whatever the shards' own k."""
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


@pytest.mark.parametrize('with_min', [False, True])
@pytest.mark.parametrize('k,kp', [(1024, 257), (1024, 513), (64, 17), (100, 101), (1280, 3), (256, 33), (512, 258)])
def test_keys_split_equals_the_restatement(k, kp, with_min):
    import torch
    from ann_solo_amd.distributed import HipShardBackend
    be = HipShardBackend.__new__(HipShardBackend)
    rng = np.random.default_rng(k + kp)
    K = np.concatenate([_rows(rng, 40, k, fill=0.5), _rows(rng, 24, k, grid=50),
                        np.zeros((2, k), np.int64)])
    head, floors = be.keys_split(torch.from_numpy(K).cuda(), kp, with_min)
    head, floors = head.cpu().numpy(), floors.cpu().numpy()
    h0, f0 = X.keys_split(K, kp, with_min)
    nkeep = kp - 1 - int(with_min)
    assert np.array_equal(head[:, nkeep:], h0[:, nkeep:])                     # (M,) best held-back key
    if with_min:                                                              # M: the smallest key of a FULL row
        Ku = K.view(np.uint64)
        full = (Ku != 0).all(1)
        assert full.any() and (~full).any()
        assert np.array_equal(head[:, kp - 2].view(np.uint64), np.where(full, Ku.min(1), 0))
    assert np.array_equal(floors, f0)                                         # the bucket floor of every row
    assert _sets(head[:, :nkeep]) == _sets(h0[:, :nkeep])
    rest = X.held_back(K, floors)                                             # what stays behind in K
    for r in range(len(K)):                                                   # zero padding is at the end
        nz = np.nonzero(head[r, :nkeep])[0]
        assert len(nz) == 0 or nz[-1] == len(nz) - 1
        a, h = _sets(head[r:r + 1, :nkeep])[0], _sets(rest[r:r + 1])[0]
        assert not h or not a or max(h) < min(a)                              # kept keys beat held-back ones
        assert len(a) <= nkeep and a | h == _sets(K[r:r + 1])[0]


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


@pytest.mark.parametrize('S,k,ks,head_keys,xper,grid,skew', [
    (8, 1024, 512, 256, 64, None, 0.45), (8, 1024, 512, 256, 1024, None, 0.9), (4, 1024, 640, 512, 1024, None, 0.6),
    (8, 1024, 128, 64, 1024, None, 0.2),        # 8 x 128 = k: nearly every full row is asked about
    (3, 256, 64, 16, 256, 30, 0.5), (8, 512, 192, 128, 512, 8, 0.5), (2, 64, 16, 8, 64, None, 0.3),
    (8, 1024, 512, 256, 2, None, 0.9)])         # answer buffers of 2 slots per query: overflow
def test_third_phase_is_the_top_k_of_the_union_of_the_full_rows(S, k, ks, head_keys, xper, grid, skew):
    """The shards hold FULL rows of k keys but send the exchange only their k_s best (what a scan
    with a shard-side k_s keeps). Heads with M -> bounds -> held-back keys -> final merge + requests
    (B', M) -> rescan answers (the keys of the full row between the two) -> last merge: every step
    equals the restatement, the result is the top k of the union of the FULL rows."""
    import torch
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
    kp = head_keys + 2
    dev_rows = [torch.from_numpy(r).cuda() for r in rows]
    split = [be.keys_split(r, kp, True) for r in dev_rows]
    heads = torch.stack([h for h, _ in split])
    out, bounds, need = be.keys_merge_heads(heads, k, True)
    o0, b0, n0 = X.keys_merge_heads(heads.cpu().numpy(), k, True)
    assert _sets(out.cpu().numpy()) == _sets(o0)
    assert np.array_equal(bounds.cpu().numpy(), b0) and np.array_equal(need.cpu().numpy(), n0)
    flag = torch.zeros(2, dtype=torch.int32, device='cuda')
    xcap = n * max(64, ks)
    xb = torch.stack([be.keys_extras(dev_rows[s], split[s][1], bounds[s].contiguous(), 1, xcap, flag)[0]
                      for s in range(S)])
    assert int(flag[0].item()) == 0
    I, fin, req, need3 = be.keys_merge_final(heads, xb, out, need, k, flag)
    I0, fin0, req0, need30, n30 = X.keys_merge_final(heads.cpu().numpy(), xb.cpu().numpy(), o0, n0, k, with_min=True)
    assert _sets(fin.cpu().numpy()) == _sets(fin0)
    assert [set(r[r >= 0].tolist()) for r in I.cpu().numpy()] == [set(r[r >= 0].tolist()) for r in I0]
    assert np.array_equal(req.cpu().numpy(), req0) and np.array_equal(need3.cpu().numpy(), need30)
    assert int(flag[1].item()) == n30
    # the union of the k_s-rows is what phases 1-2 must have produced
    u12 = np.concatenate(rows, axis=1).view(np.uint64)
    assert _sets(fin.cpu().numpy()) == [set(np.sort(u[u != 0])[::-1][:k].tolist()) for u in u12]
    # phase 3: every shard answers the one owner (world = 1 on the shard side)
    flag.zero_()
    xcap3 = n * xper
    answers, over_ref = [], 0
    for s in range(S):
        rq = req[s].contiguous()                                 # [n, 2]
        sel = be.request_rows(rq)
        assert np.array_equal(sel.cpu().numpy(), X.request_rows(req0[s]))
        K3 = torch.from_numpy(full[s]).cuda().index_select(0, sel)
        answers.append(be.keys_rescan(K3, sel, rq, 1, n, xcap3, flag)[0])
        x0, ov = X.keys_rescan(full[s][sel.cpu().numpy()], sel.cpu().numpy(), req0[s], 1, n, xcap3)
        over_ref |= ov
        if not ov:
            got, exp = answers[-1].cpu().numpy().view(np.uint64), x0[0].view(np.uint64)
            for q in range(n):
                c, st = int(got[q] >> np.uint64(32)), int(got[q] & np.uint64(0xFFFFFFFF))
                c0, st0 = int(exp[q] >> np.uint64(32)), int(exp[q] & np.uint64(0xFFFFFFFF))
                assert c == c0 and set(got[n + st:n + st + c].tolist()) == set(exp[n + st0:n + st0 + c0].tolist())
    assert int(flag[0].item()) == over_ref
    if over_ref:
        assert xper < k
        return
    I3 = be.keys_merge3(fin, torch.stack(answers), need3, k).cpu().numpy()
    union = np.concatenate(full, axis=1).view(np.uint64)
    want = [set(X.key_id(np.sort(u[u != 0])[::-1][:k]).tolist()) for u in union]
    asked = 0
    for q in range(n):
        got = I3[q][I3[q] >= 0]
        assert len(got) == len(set(got.tolist())) == len(want[q]) and set(got.tolist()) == want[q], q
        assert (I3[q][len(got):] == -1).all()
        asked += int(need30[q])
    assert asked > 0 or ks * S > k               # rows that cannot fill k between them: everybody asks

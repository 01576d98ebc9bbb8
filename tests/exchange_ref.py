"""numpy restatement of the two-phase exact top-k exchange (ann_solo_amd/csrc/exchange.hip) --
test infrastructure: the CPU backend of the gloo tests (tests/oracle_backend.py) and the checker
of the device kernels (tests/test_gpu_exchange.py). No reference counterpart: the reference has
no multi-GPU code (/root/reference/src/ann_solo/spectral_library.py:494).

Keys: uint64 (order-preserving bits of the fp32 score) << 32 | (0xFFFFFFFF - id); 0 = empty;
larger key = better hit under (score desc, id asc). Rows are int64 arrays holding those bits."""
import numpy as np

NONE = np.uint64(0xFFFFFFFFFFFFFFFF)
HT_NB = 512


def f2ord(score):
    u = np.asarray(score, np.float32).view(np.uint32)
    return np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def ord2f(o):
    o = np.asarray(o, np.uint32)
    u = np.where(o & np.uint32(0x80000000), o & np.uint32(0x7FFFFFFF), ~o).astype(np.uint32)
    return u.view(np.float32)


def pack_keys(D, I):
    """(D float32, I int64 with -1 padding) -> packed keys (int64 bit patterns), 0 where I < 0."""
    D, I = np.asarray(D, np.float32), np.asarray(I, np.int64)
    key = (f2ord(D).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - np.maximum(I, 0).astype(np.uint64))
    return np.where(I >= 0, key, np.uint64(0)).astype(np.uint64).view(np.int64)


def key_id(K):
    K = np.asarray(K).view(np.uint64)
    return np.where(K != 0, (np.uint64(0xFFFFFFFF) - (K & np.uint64(0xFFFFFFFF))).astype(np.int64), -1)


def key_score(K):
    return ord2f((np.asarray(K).view(np.uint64) >> np.uint64(32)).astype(np.uint32))


def score_bucket(s):
    """hist_topk.hpp: monotone linear bucketing of the fp32 score over [-0.25, 1), float32 steps."""
    s = np.asarray(s, np.float32)
    t = (s - np.float32(-0.25)) * (np.float32(HT_NB) / np.float32(1.25))
    t = np.minimum(np.maximum(t, np.float32(0)), np.float32(HT_NB - 1))
    return t.astype(np.int32)


def shard_k(k, world):
    """asl_shard_k (csrc/index.hip): the shards' own k."""
    if k < 1 or world < 4:
        return k
    raw = (k + 1) // 2 if world >= 8 else (5 * k + 7) // 8
    ks = min(k, (raw + 63) // 64 * 64)
    head = min(k, -(-2 * k // world))
    return ks if ks > head else k


def row_min(K):
    """M of every row: its smallest key if the row is FULL (no empty slot), else 0."""
    K = np.asarray(K).view(np.uint64)
    return np.where((K != 0).all(1), K.min(1), np.uint64(0)).astype(np.uint64).view(np.int64)


def keys_split(K, kp):
    """K [rows, k] -> head [rows, kp] (kept keys, then 0; last slot = best held-back key), floor
    [rows] (int32). Kept = every key whose bucket is at or above the lowest bucket floor that admits
    at most kp - 1 keys; the held-back keys are the keys of K below the floor (``held_back``)."""
    K = np.asarray(K).view(np.uint64)
    rows, k = K.shape
    head = np.zeros((rows, kp), np.uint64)
    floors = np.zeros(rows, np.int32)
    for r in range(rows):
        keys = K[r][K[r] != 0]
        b = score_bucket(key_score(keys))
        cum = np.cumsum(np.bincount(b, minlength=HT_NB)[::-1])[::-1]      # keys in buckets >= j
        ok = np.nonzero(cum <= kp - 1)[0]
        floor = int(ok[0]) if len(ok) else HT_NB
        a, h = keys[b >= floor], keys[b < floor]
        head[r, :len(a)] = a
        head[r, kp - 1] = h.max() if len(h) else 0
        floors[r] = floor
    return head.view(np.int64), floors


def held_back(K, floors):
    """The keys of every row of K below the row's bucket floor, compacted and 0 padded [rows, k]
    (what the split holds back; test helper)."""
    K = np.asarray(K).view(np.uint64)
    rest = np.zeros_like(K)
    for r in range(len(K)):
        keys = K[r][K[r] != 0]
        h = keys[score_bucket(key_score(keys)) < floors[r]]
        rest[r, :len(h)] = h
    return rest.view(np.int64)


def _topk_set(cands, k):
    cands = np.unique(cands[cands != 0])            # ascending, keys are unique anyway
    return cands[::-1][:k]


def keys_merge_heads(heads, k):
    """heads [S, n, kp] -> out_keys [n, k] (best k keys seen, descending here; the device emits a
    set), bounds [S, n] (B or NONE), need [n]."""
    heads = np.asarray(heads).view(np.uint64)
    S, n, kp = heads.shape
    out = np.zeros((n, k), np.uint64)
    bounds = np.full((S, n), NONE, np.uint64)
    need = np.zeros(n, np.int32)
    for q in range(n):
        best = _topk_set(heads[:, q, :kp - 1].reshape(-1), k)
        out[q, :len(best)] = best
        B = best[-1] if len(best) >= k else np.uint64(0)
        ask = heads[:, q, kp - 1] > B
        bounds[ask, q] = B
        need[q] = int(ask.any())
    return out.view(np.int64), bounds.view(np.int64), need


def rescan_list(bounds, rowmin, R):
    """Rows whose bound lies below the smallest key of a full row (a dropped key may be above the
    bound): -> rowlist [R] (the first count slots, ascending here; the device's order is whatever
    its atomics gave), rmap [rows] (slot or -1), count, overflow (more than R rows)."""
    bounds = np.asarray(bounds).view(np.uint64).reshape(-1)
    rowmin = np.asarray(rowmin).view(np.uint64).reshape(-1)
    need = np.nonzero((bounds != NONE) & (rowmin != 0) & (rowmin > bounds))[0]
    rowlist = np.zeros(R, np.int64)
    rmap = np.full(len(bounds), -1, np.int32)
    take = need[:R]
    rowlist[:len(take)] = take
    rmap[take] = np.arange(len(take), dtype=np.int32)
    return rowlist, rmap, len(need), int(len(need) > R)


def keys_extras(K, floors, bounds, world, xcap, rmap=None, K3=None):
    """K [world * n, k] (destination-major rows) and the floors of ``keys_split``, bounds [world * n] -> xbuf [world, n + xcap]
    (n header words count << 32 | start, then the payload) and the overflow flag. Payload order
    inside a destination is the row order here (the device's is whatever its atomics gave: the
    headers say where each row's keys are). ``rmap`` / ``K3``: rows with rmap[row] >= 0 answer from
    K3[rmap[row]] (their second scan with the full k: every key below the floor counts)."""
    rest = held_back(K, floors).view(np.uint64)
    if rmap is not None:
        K3u = np.asarray(K3).view(np.uint64)
        wide = np.zeros((len(rest), max(rest.shape[1], K3u.shape[1])), np.uint64)
        wide[:, :rest.shape[1]] = rest
        for r in np.nonzero(np.asarray(rmap) >= 0)[0]:
            row3 = K3u[rmap[r]]
            keep = row3[(row3 != 0) & (score_bucket(key_score(row3)) < floors[r])]
            wide[r] = 0
            wide[r, :len(keep)] = keep
        rest = wide
    bounds = np.asarray(bounds).view(np.uint64)
    rows, k = rest.shape
    n = rows // world
    xbuf = np.zeros((world, n + xcap), np.uint64)
    overflow = 0
    for d in range(world):
        cur = 0
        for q in range(n):
            B = bounds[d * n + q]
            if B == NONE:
                continue
            r = rest[d * n + q]
            e = r[(r != 0) & (r > B)]
            if cur + len(e) > xcap:
                overflow = 1
                cur += len(e)
                continue
            xbuf[d, q] = (np.uint64(len(e)) << np.uint64(32)) | np.uint64(cur)
            xbuf[d, n + cur:n + cur + len(e)] = e
            cur += len(e)
    return xbuf.view(np.int64), overflow


def keys_merge_final(heads, xbuf, out_keys, need, k):
    """-> I [n, k] ids of the exact top-k of everything received (descending here), -1 padded."""
    heads = np.asarray(heads).view(np.uint64)
    S, n, kp = heads.shape
    I = np.full((n, k), -1, np.int64)
    xb = None if xbuf is None else np.asarray(xbuf).view(np.uint64)
    for q in range(n):
        if xb is None or not need[q]:
            keys = np.asarray(out_keys).view(np.uint64)[q]
            keys = keys[keys != 0]
        else:
            parts = [heads[:, q, :kp - 1].reshape(-1)]
            for s_ in range(S):
                h = xb[s_, q]
                cnt, st = int(h >> np.uint64(32)), int(h & np.uint64(0xFFFFFFFF))
                parts.append(xb[s_, n + st:n + st + cnt])
            keys = _topk_set(np.concatenate(parts), k)
        I[q, :len(keys)] = key_id(keys)
    return I

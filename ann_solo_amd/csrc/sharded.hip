// sharded.hip -- the list-sharded IVF search behind the C ABI, over an RCCL communicator the
// CALLER owns (asl_index_shard(idx, rank, world, rcclComm)). The Python
// engine drives the same exchange through torch.distributed (ann_solo_amd/distributed.py); this
// entry point is for hosts that bind the C ABI directly.
//
// RCCL is not linked: the ncclXxx entry points are resolved at first use from the process
// (a host that created a communicator has RCCL loaded already -- e.g. PyTorch's bundled copy) and
// only then from librccl.so.1, so the library never drags a second RCCL into the process.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "ivf_kernels.hpp"

struct asl_index;

namespace asl {
int index_search_device(asl_index *ix, int nq, const float *xq, int k, int nprobe, float *D,
                        int64_t *I64, int32_t *I32, const float *pre_D, const int32_t *pre_I,
                        bool set_mode, const int *gate, const uint2 *pre_ent = nullptr,
                        const int32_t *pre_cnt = nullptr);
int index_dim(const asl_index *ix);
int index_nprobe(const asl_index *ix, int nprobe);
int index_prepare(asl_index *ix);
int index_coarse_device(asl_index *ix, int nq, const float *xq, int nprobe, float *out_D,
                        int32_t *out_I, uint2 *ent_out = nullptr, int32_t *cnt_out = nullptr,
                        bool *have_ent = nullptr);
int index_shard_world(const asl_index *ix, int *rank);
int index_agreed_keys(const asl_index *ix, int k, int np, int world);
void index_set_agreed_keys(asl_index *ix, int k, int np, int world, int v);
int index_refine_k(const asl_index *ix);
int index_swap_unordered(asl_index *ix, int mode, int *prev);
int index_refine_device(asl_index *ix, int nq, const float *xq, int kp, const int64_t *I_in, int k,
                        float *D, int64_t *I);

// the few RCCL declarations used (rccl.h: stable since NCCL 2.7)
typedef void *nccl_comm_t;
typedef int nccl_result_t;
constexpr int NCCL_INT32 = 2, NCCL_INT64 = 4, NCCL_FLOAT32 = 7;
struct Rccl {
  nccl_result_t (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  nccl_result_t (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  nccl_result_t (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  nccl_result_t (*GroupStart)() = nullptr;
  nccl_result_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(nccl_result_t) = nullptr;
  bool ok = false;
};

static Rccl &rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  // 1. ASL_RCCL_LIB (explicit path); 2. whatever RCCL the process exposes globally; 3. an
  // instance that is loaded already under the usual soname; 4. a fresh load
  void *h = nullptr;
  if (const char *path = getenv("ASL_RCCL_LIB")) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h && dlsym(RTLD_DEFAULT, "ncclAllGather")) h = RTLD_DEFAULT;
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return r;
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
  r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(h, "ncclSend"));
  r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(h, "ncclRecv"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(h, "ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  r.ok = r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
  return r;
}

#define RCCL_TRY(expr)                                                                        \
  do {                                                                                        \
    nccl_result_t r_ = (expr);                                                                \
    if (r_ != 0)                                                                              \
      return fail(ASL_ERR_HIP, "%s: %s", #expr, R.GetErrorString ? R.GetErrorString(r_) : "RCCL error"); \
  } while (0)

}  // namespace asl

using namespace asl;

// Every rank calls this with ITS nq queries (the same nq on every rank, device pointers) after
// asl_index_shard(idx, rank, world) on an index holding the same vectors everywhere:
//   all-gather of the hashed queries (as entry lists -- their <= 64 non-zero components, 516 bytes
//   instead of 3.2 KB per query -- whenever the packed-key scans run, which read them in that
//   form) -> coarse quantiser on the own slice, probe lists
//   all-gathered -> scan of the local inverted lists for all world x nq queries (exact top-k
//   sets) -> the two-phase exact exchange of exchange.hip (heads of ~2k / world packed keys,
//   the owners' bounds, the held-back keys above them; grouped send/recv = direct peer copies
//   over xGMI) -> merge under (score desc, id asc). The full rows travel instead when the scan
//   cannot emit packed keys, with the exact re-rank on, or after a phase-2 buffer ran full.
// With the exact re-rank on (asl_index_set_refine, k' > k) every shard returns its k' best ADC
// hits, the merge yields the k' best of the whole index -- the unsharded short-list -- and the
// owner of the query re-ranks that against the exact rows (replicated on every rank).
// D / I [nq, k]: identical to what the unsharded index returns for these queries.
// head_keys / shard_keys / extras_per_query: the exchange's three sizes (keys a head carries, the
// shards' own k, answer slots per query and destination); <= 0 / <= 0 / < 0 = the defaults
// (ceil(2k / world), asl_shard_k(k, world), max(8, k / 16)). Results do not depend on them; tests
// set them to run the bound / held-back-key / second-scan steps and the overflow fallback on a
// communicator of any size (at world 1 the default head is the whole row).
extern "C" int asl_index_search_sharded_ex(asl_index_t *ix, void *rccl_comm, int32_t nq,
                                           const float *xq, int32_t k, int32_t nprobe, float *D,
                                           int64_t *I, int32_t head_keys, int32_t shard_keys,
                                           int64_t extras_per_query) {
  clear_error();
  if (!ix || !rccl_comm || !xq || !I) return fail(ASL_ERR_INVALID, "search_sharded: null argument");
  if (nq <= 0) return fail(ASL_ERR_INVALID, "search_sharded: every rank must bring nq > 0 queries");
  if (k <= 0 || k > TK_MAX_K) return fail(ASL_ERR_CAPACITY, "search_sharded: k=%d outside 1..%d", k, TK_MAX_K);
  ASL_TRY(ensure_device());
  int rank = 0;
  const int world = index_shard_world(ix, &rank);
  if (world < 1) return fail(ASL_ERR_STATE, "search_sharded: call asl_index_shard first");
  if (!is_device_ptr(xq) || !is_device_ptr(I) || (D && !is_device_ptr(D)))
    return fail(ASL_ERR_INVALID, "search_sharded: queries and results must be device memory");
  Rccl &R = rccl();
  if (!R.ok) return fail(ASL_ERR_STATE, "search_sharded: RCCL (librccl.so.1) is not available in this process");
  const int d = index_dim(ix);
  const int np = index_nprobe(ix, nprobe);
  if (np <= 0) return fail(ASL_ERR_INVALID, "search_sharded: an IVF index is required");
  ASL_TRY(index_prepare(ix));
  const int k_out = k;
  const int kref = index_refine_k(ix);
  const bool refine = kref > k;
  if (refine) k = std::min(kref, (int)TK_MAX_K);     // per-shard rows and the merge carry k' hits
  const size_t all = (size_t)world * nq;
  static DevBuf<float> &x_all = *new DevBuf<float>(), &cD = *new DevBuf<float>(),
                       &cD_all = *new DevBuf<float>(), &Dp = *new DevBuf<float>(),
                       &Dr = *new DevBuf<float>(), &Dtmp = *new DevBuf<float>();
  static DevBuf<int32_t> &cI = *new DevBuf<int32_t>(), &cI_all = *new DevBuf<int32_t>();
  static DevBuf<int64_t> &Ip = *new DevBuf<int64_t>(), &Ir = *new DevBuf<int64_t>();
  // the queries as entry lists (coarse_sparse.hip: list_nonzeros): own, everybody's, second scans
  static DevBuf<uint2> &e_loc = *new DevBuf<uint2>(), &e_all = *new DevBuf<uint2>(), &e3 = *new DevBuf<uint2>();
  static DevBuf<int32_t> &c_loc = *new DevBuf<int32_t>(), &c_all = *new DevBuf<int32_t>(), &c3 = *new DevBuf<int32_t>(),
                         &over = *new DevBuf<int32_t>();
  ASL_TRY(cD.reserve((size_t)nq * np));
  ASL_TRY(cI.reserve((size_t)nq * np));
  ASL_TRY(cD_all.reserve(all * np));
  ASL_TRY(cI_all.reserve(all * np));
  hipStream_t st = stream();
  nccl_comm_t comm = rccl_comm;
  // a few ints every rank must see the same way: gathered, downloaded (one stream synchronisation)
  static DevBuf<int32_t> &mine = *new DevBuf<int32_t>(), &everyone = *new DevBuf<int32_t>();
  ASL_TRY(mine.reserve(4));
  ASL_TRY(everyone.reserve((size_t)world * 4));
  auto agree = [&](std::vector<int32_t> &h) -> int {      // mine[0..3] of every rank -> h [world * 4]
    h.assign((size_t)world * 4, 0);
    RCCL_TRY(R.AllGather(mine.p, everyone.p, 4, NCCL_INT32, comm, st));
    ASL_TRY(everyone.download(h.data(), (size_t)world * 4));
    return sync_stream();
  };
  std::vector<int32_t> h;
  // The exact key exchange (exchange.hip) whenever EVERY shard's scan can emit packed keys (for
  // IVF-Flat that depends on the vectors a shard holds: agreed across the ranks, once per state
  // of the index): heads of ~2k / world keys, the owners' bounds, the held-back keys above
  // them (the shards scanning with k_s < k and a second, full-k scan where a bound asks); the full
  // rows remain the path for everything else and the fallback when an answer buffer runs full
  bool keys_everywhere = false;
  if (!refine) {
    // (asl_index_supports_keys first: it rebuilds a stale scan layout, which forgets the agreement)
    const int32_t local = asl_index_supports_keys(ix, k, np) ? 1 : 0;
    int known = index_agreed_keys(ix, k, np, world);
    // every rank must take the same branch HERE too: the agreement is made on the first call after
    // any change of the index (all ranks change it together: add / shard are collective by contract)
    if (known < 0) {
      int32_t hm[4] = {local, 0, 0, 0};
      ASL_TRY(mine.upload(hm, 4));
      ASL_TRY(agree(h));
      known = 1;
      for (int r = 0; r < world; ++r) known = known && h[(size_t)r * 4] != 0;
      index_set_agreed_keys(ix, k, np, world, known);
    }
    keys_everywhere = known != 0;
  }
  // The packed-key scans read their queries as ENTRY LISTS -- the non-zero components, at most 64
  // (dimension * 128, value bits) -- so that is the form the queries travel in: 516 bytes per
  // query on the wire instead of 3.2 KB, no dense row per (rank, query) in memory. A query with
  // more non-zeros has no such form: any rank that meets one says so in the step's agreement and
  // the batch is repeated with dense rows (`entries` = false).
  bool redo_dense = false;
  auto run = [&](const bool entries) -> int {
    // 1. everybody's queries; the coarse quantiser runs on the own slice meanwhile (same stream:
    //    RCCL orders itself after it; a second stream would overlap the two)
    if (entries) {
      ASL_TRY(e_loc.reserve((size_t)nq * 64));
      ASL_TRY(c_loc.reserve((size_t)nq));
      ASL_TRY(e_all.reserve(all * 64));
      ASL_TRY(c_all.reserve(all));
      ASL_TRY(over.reserve(1));
      HIP_TRY(hipMemsetAsync(over.p, 0, sizeof(int32_t), st));
      ASL_TRY(list_nonzeros(xq, nq, d, d, e_loc.p, c_loc.p, over.p));
      RCCL_TRY(R.AllGather(e_loc.p, e_all.p, (size_t)nq * 128, NCCL_INT32, comm, st));
      RCCL_TRY(R.AllGather(c_loc.p, c_all.p, (size_t)nq, NCCL_INT32, comm, st));
    } else {
      ASL_TRY(x_all.reserve(all * d));
      RCCL_TRY(R.AllGather(xq, x_all.p, (size_t)nq * d, NCCL_FLOAT32, comm, st));
    }
    const float *xs = entries ? nullptr : x_all.p;
    const uint2 *es = entries ? e_all.p : nullptr;
    const int32_t *cs = entries ? c_all.p : nullptr;
    ASL_TRY(index_coarse_device(ix, nq, xq, np, cD.p, cI.p));
    RCCL_TRY(R.AllGather(cD.p, cD_all.p, (size_t)nq * np, NCCL_FLOAT32, comm, st));
    RCCL_TRY(R.AllGather(cI.p, cI_all.p, (size_t)nq * np, NCCL_INT32, comm, st));
    if (keys_everywhere) {
      typedef unsigned long long u64k;
      static DevBuf<int64_t> &Kp = *new DevBuf<int64_t>(), &Hs = *new DevBuf<int64_t>(), &Hr = *new DevBuf<int64_t>(),
                             &Ko = *new DevBuf<int64_t>(), &Bs = *new DevBuf<int64_t>(),
                             &Br = *new DevBuf<int64_t>(), &Xs = *new DevBuf<int64_t>(), &Xr = *new DevBuf<int64_t>(),
                             &Mn = *new DevBuf<int64_t>(), &rowlist = *new DevBuf<int64_t>(), &K3 = *new DevBuf<int64_t>();
      static DevBuf<int32_t> &need = *new DevBuf<int32_t>(), &flag = *new DevBuf<int32_t>(), &Fl = *new DevBuf<int32_t>(),
                             &rmap = *new DevBuf<int32_t>(), &cI3 = *new DevBuf<int32_t>();
      static DevBuf<float> &x3 = *new DevBuf<float>(), &cD3 = *new DevBuf<float>();
      static DevBuf<unsigned int> &cursor = *new DevBuf<unsigned int>();
      const int keys = std::min(k, head_keys > 0 ? (int)head_keys : (2 * k + world - 1) / world), kp = keys + 1;
      const bool second = keys < k;                          // heads hold something back
      int ks = second ? (shard_keys > 0 ? (int)shard_keys : asl_shard_k(k, world)) : k;   // the shards' own k (exchange.hip)
      if (!(keys < ks && ks < k)) ks = k;
      const bool rescan = ks < k;
      const long long xcap = (long long)nq * (extras_per_query >= 0 ? (long long)extras_per_query : std::max(8, k / 16));
      ASL_TRY(Kp.reserve(all * ks));
      ASL_TRY(Hs.reserve(all * kp));
      ASL_TRY(Hr.reserve(all * kp));
      ASL_TRY(Fl.reserve(all));
      ASL_TRY(Ko.reserve((size_t)nq * k));
      ASL_TRY(Bs.reserve(all));
      ASL_TRY(Br.reserve(all));
      ASL_TRY(need.reserve((size_t)nq));
      ASL_TRY(flag.reserve(2));          // [0] a buffer ran full, [1] rows this shard scans a second time
      if (rescan) ASL_TRY(Mn.reserve(all));
      HIP_TRY(hipMemsetAsync(flag.p, 0, 2 * sizeof(int32_t), st));
      int prev = 0;
      ASL_TRY(index_swap_unordered(ix, 2, &prev));
      const int rc = index_search_device(ix, (int)all, xs, ks, np, nullptr, Kp.p, nullptr, cD_all.p, cI_all.p, true,
                                         nullptr, es, cs);
      ASL_TRY(index_swap_unordered(ix, prev, nullptr));
      ASL_TRY(rc);
      ASL_TRY(keys_split(reinterpret_cast<const u64k *>(Kp.p), (int64_t)all, ks, kp, reinterpret_cast<u64k *>(Hs.p),
                         Fl.p, rescan ? reinterpret_cast<u64k *>(Mn.p) : nullptr));
      auto all_to_all = [&](const int64_t *src, int64_t *dst, size_t per_rank) -> int {
        RCCL_TRY(R.GroupStart());
        for (int r = 0; r < world; ++r) {
          RCCL_TRY(R.Send(src + (size_t)r * per_rank, per_rank, NCCL_INT64, r, comm, st));
          RCCL_TRY(R.Recv(dst + (size_t)r * per_rank, per_rank, NCCL_INT64, r, comm, st));
        }
        RCCL_TRY(R.GroupEnd());
        return ASL_OK;
      };
      ASL_TRY(all_to_all(Hs.p, Hr.p, (size_t)nq * kp));
      ASL_TRY(keys_merge(reinterpret_cast<const u64k *>(Hr.p), world, nq, kp, k, nullptr, 0, nullptr, need.p,
                         reinterpret_cast<u64k *>(Ko.p), reinterpret_cast<u64k *>(Bs.p), nullptr, nullptr, 0));
      bool overflow = false;
      const int64_t *xr = nullptr;
      if (second) {
        ASL_TRY(Xs.reserve((size_t)world * ((size_t)nq + (size_t)xcap)));
        ASL_TRY(Xr.reserve((size_t)world * ((size_t)nq + (size_t)xcap)));
        ASL_TRY(cursor.reserve((size_t)world));
        ASL_TRY(all_to_all(Bs.p, Br.p, (size_t)nq));
        HIP_TRY(hipMemsetAsync(cursor.p, 0, (size_t)world * sizeof(unsigned int), st));
        const int32_t *rm = nullptr;
        if (rescan) {
          // rows whose bound lies below the smallest key of a full k_s-row are scanned again with the
          // full k: a launch of `cap` workgroups gated by the device-side count (no host round trip)
          const int cap = (int)std::max<size_t>(64, all / 16);
          ASL_TRY(rowlist.reserve((size_t)cap));
          ASL_TRY(rmap.reserve(all));
          ASL_TRY(cD3.reserve((size_t)cap * np));
          ASL_TRY(cI3.reserve((size_t)cap * np));
          ASL_TRY(K3.reserve((size_t)cap * k));
          HIP_TRY(hipMemsetAsync(rowlist.p, 0, (size_t)cap * sizeof(int64_t), st));
          ASL_TRY(rescan_list(reinterpret_cast<const u64k *>(Br.p), reinterpret_cast<const u64k *>(Mn.p), (int64_t)all, cap,
                              rowlist.p, rmap.p, reinterpret_cast<int *>(flag.p + 1), flag.p));
          if (entries) {               // (rows of 4-byte words)
            ASL_TRY(e3.reserve((size_t)cap * 64));
            ASL_TRY(c3.reserve((size_t)cap));
            ASL_TRY(gather_rows_f32(reinterpret_cast<const float *>(e_all.p), 128, rowlist.p, cap, 128,
                                    reinterpret_cast<float *>(e3.p), 128));
            ASL_TRY(gather_rows_f32(reinterpret_cast<const float *>(c_all.p), 1, rowlist.p, cap, 1,
                                    reinterpret_cast<float *>(c3.p), 1));
          } else {
            ASL_TRY(x3.reserve((size_t)cap * d));
            ASL_TRY(gather_rows_f32(x_all.p, d, rowlist.p, cap, d, x3.p, d));
          }
          ASL_TRY(gather_rows_f32(cD_all.p, np, rowlist.p, cap, np, cD3.p, np));
          ASL_TRY(gather_rows_f32(reinterpret_cast<const float *>(cI_all.p), np, rowlist.p, cap, np,
                                  reinterpret_cast<float *>(cI3.p), np));      // (4-byte words)
          ASL_TRY(index_swap_unordered(ix, 2, &prev));
          const int rc3 = index_search_device(ix, cap, entries ? nullptr : x3.p, k, np, nullptr, K3.p, nullptr, cD3.p,
                                              cI3.p, true, reinterpret_cast<const int *>(flag.p + 1),
                                              entries ? e3.p : nullptr, entries ? c3.p : nullptr);
          ASL_TRY(index_swap_unordered(ix, prev, nullptr));
          ASL_TRY(rc3);
          rm = rmap.p;
        }
        ASL_TRY(keys_extras(reinterpret_cast<const u64k *>(Kp.p), Fl.p, (int64_t)all, ks, reinterpret_cast<const u64k *>(Br.p),
                            nq, xcap, reinterpret_cast<u64k *>(Xs.p), cursor.p, flag.p, rm,
                            reinterpret_cast<const u64k *>(K3.p), k));
        ASL_TRY(all_to_all(Xs.p, Xr.p, (size_t)nq + (size_t)xcap));
        xr = Xr.p;
      }
      if (second || entries) {
        // a full buffer ANYWHERE sends every rank down the full exchange, a query without an entry
        // list ANYWHERE sends them back to dense rows: all ranks take the same branch
        HIP_TRY(hipMemcpyAsync(mine.p, flag.p, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        if (entries)
          HIP_TRY(hipMemcpyAsync(mine.p + 2, over.p, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        else
          HIP_TRY(hipMemsetAsync(mine.p + 2, 0, sizeof(int32_t), st));
        ASL_TRY(agree(h));
        for (int r = 0; r < world; ++r) {
          overflow |= second && h[(size_t)r * 4] != 0;
          redo_dense |= h[(size_t)r * 4 + 2] != 0;
        }
        if (redo_dense) return ASL_OK;
      }
      if (!overflow)
        return keys_merge(reinterpret_cast<const u64k *>(Hr.p), world, nq, kp, k, reinterpret_cast<const u64k *>(xr),
                          xr ? xcap : 0, reinterpret_cast<const u64k *>(Ko.p), need.p, nullptr, nullptr, I, D, 1);
    }
    // 2. the local lists, for all queries (rank-major rows), as exact top-k sets
    //    (unordered mode: un-refined ADC rows, see annsolo_mi.h at asl_index_set_refine)
    ASL_TRY(Dp.reserve(all * k));
    ASL_TRY(Ip.reserve(all * k));
    ASL_TRY(Dr.reserve(all * k));
    ASL_TRY(Ir.reserve(all * k));
    int prev_unordered = 0;
    ASL_TRY(index_swap_unordered(ix, 1, &prev_unordered));
    const int rc_scan = index_search_device(ix, (int)all, xs, k, np, Dp.p, Ip.p, nullptr, cD_all.p, cI_all.p, true,
                                            nullptr, es, cs);
    ASL_TRY(index_swap_unordered(ix, prev_unordered, nullptr));
    ASL_TRY(rc_scan);
    // 3. rank r receives the `world` partial rows of its own queries
    RCCL_TRY(R.GroupStart());
    for (int r = 0; r < world; ++r) {
      RCCL_TRY(R.Send(Dp.p + (size_t)r * nq * k, (size_t)nq * k, NCCL_FLOAT32, r, comm, st));
      RCCL_TRY(R.Recv(Dr.p + (size_t)r * nq * k, (size_t)nq * k, NCCL_FLOAT32, r, comm, st));
      RCCL_TRY(R.Send(Ip.p + (size_t)r * nq * k, (size_t)nq * k, NCCL_INT64, r, comm, st));
      RCCL_TRY(R.Recv(Ir.p + (size_t)r * nq * k, (size_t)nq * k, NCCL_INT64, r, comm, st));
    }
    RCCL_TRY(R.GroupEnd());
    // 4. merge (and the exact re-rank of the merged short-list)
    static DevBuf<int64_t> &Im = *new DevBuf<int64_t>();
    float *Dout = D;
    if (!Dout || refine) {
      ASL_TRY(Dtmp.reserve((size_t)nq * k));
      Dout = Dtmp.p;
    }
    if (!refine) return topk_merge(Dr.p, Ir.p, world, nq, k, Dout, I);
    ASL_TRY(Im.reserve((size_t)nq * k));
    ASL_TRY(topk_merge(Dr.p, Ir.p, world, nq, k, Dout, Im.p));
    static DevBuf<float> &Dfin = *new DevBuf<float>();
    float *Df = D;
    if (!Df) {
      ASL_TRY(Dfin.reserve((size_t)nq * k_out));
      Df = Dfin.p;
    }
    return index_refine_device(ix, nq, xq, k, Im.p, k_out, Df, I);
  };
  ASL_TRY(run(keys_everywhere));
  if (redo_dense) {
    redo_dense = false;
    return run(false);
  }
  return ASL_OK;
}

extern "C" int asl_index_search_sharded(asl_index_t *ix, void *rccl_comm, int32_t nq,
                                        const float *xq, int32_t k, int32_t nprobe, float *D,
                                        int64_t *I) {
  return asl_index_search_sharded_ex(ix, rccl_comm, nq, xq, k, nprobe, D, I, 0, 0, -1);
}

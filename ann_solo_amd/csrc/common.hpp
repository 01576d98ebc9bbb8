// common.hpp -- shared host-side plumbing of libannsolo_mi.so (errors, stream,
// host/device staging, HIP-event stage timers). gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/annsolo_mi.h"

namespace asl {

int fail(int code, const char *fmt, ...);
void clear_error();

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return ::asl::fail(ASL_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                         __FILE__, __LINE__);                                              \
  } while (0)

#define ASL_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != ASL_OK) return rc_; \
  } while (0)

#define ASL_CHECK_LAUNCH() HIP_TRY(hipGetLastError())

hipStream_t stream();
int ensure_device();  // ASL_OK if a HIP device is usable
bool is_device_ptr(const void *p);

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
// An AQL dispatch packet holds a dimension's grid size in WORK-ITEMS as a uint32: a launch of 2^32
// or more work-items along x runs cut short, with no error (found at 134 M PQ codes: 4.3e9
// per-byte threads -- profiles/r06_pq_beyond_llc_notes.txt). Launches whose size grows with the
// library take their blocks from this 2-D grid and read their index through block_linear().
inline dim3 grid_2d(int64_t blocks) {
  const int64_t gx = std::min<int64_t>(std::max<int64_t>(blocks, 1), 1 << 20);
  return dim3((unsigned)gx, (unsigned)cdiv(std::max<int64_t>(blocks, 1), gx));
}
__device__ __forceinline__ int64_t block_linear() { return (int64_t)blockIdx.y * gridDim.x + blockIdx.x; }

// Grow-only device buffer.
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      cap = o.cap;
      o.p = nullptr;
      o.cap = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  int reserve(size_t n) {
    if (n <= cap) return ASL_OK;
    release();
    if (n == 0) return ASL_OK;
    HIP_TRY(hipMalloc((void **)&p, n * sizeof(T)));
    cap = n;
    return ASL_OK;
  }
  int upload(const T *src, size_t n) {  // src host or device
    ASL_TRY(reserve(n));
    if (n) HIP_TRY(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyDefault, stream()));
    return ASL_OK;
  }
  int download(T *dst, size_t n) const {
    if (n) HIP_TRY(hipMemcpyAsync(dst, p, n * sizeof(T), hipMemcpyDefault, stream()));
    return ASL_OK;
  }
};

// Read-only argument that may live on the host: gives a device pointer.
template <class T>
struct In {
  const T *d = nullptr;
  DevBuf<T> own;
  int init(const T *src, size_t n) {
    if (src == nullptr || n == 0) {
      d = nullptr;
      return ASL_OK;
    }
    if (is_device_ptr(src)) {
      d = src;
      return ASL_OK;
    }
    ASL_TRY(own.upload(src, n));
    d = own.p;
    return ASL_OK;
  }
};

// Output argument that may live on the host: kernels write to .d, finish() copies back.
template <class T>
struct Out {
  T *d = nullptr;
  T *host = nullptr;
  size_t n = 0;
  DevBuf<T> own;
  int init(T *dst, size_t count) {
    n = count;
    host = nullptr;
    if (dst == nullptr || count == 0) {
      d = nullptr;
      return ASL_OK;
    }
    if (is_device_ptr(dst)) {
      d = dst;
      return ASL_OK;
    }
    ASL_TRY(own.reserve(count));
    d = own.p;
    host = dst;
    return ASL_OK;
  }
  bool to_host() const { return host != nullptr; }
  int finish() {  // enqueue the copy-back (caller synchronises once at the end)
    if (host && n) HIP_TRY(hipMemcpyAsync(host, d, n * sizeof(T), hipMemcpyDeviceToHost, stream()));
    return ASL_OK;
  }
};

int sync_stream();

// Work issued inside the scope goes to `s` (every kernel launch reads stream()).
struct StreamScope {
  hipStream_t prev;
  explicit StreamScope(hipStream_t s);
  ~StreamScope();
};

// Software pipeline of asl_search_batch over three streams (asl_set_pipeline): A runs the
// MFMA-bound front of batch i+2 (encode, coarse GEMM, coarse select), B the list scan of batch
// i+1, C the filter + rescoring of batch i. Buffers that cross streams exist twice (by parity).
struct Pipeline {
  bool on = false, inflight = false, in_call = false;
  int streams = 2;   // 2: front | scan + rescoring; 3: front | scan | rescoring
  hipStream_t A = nullptr, B = nullptr, C = nullptr;
  hipEvent_t ev_in = nullptr, ev_front[2] = {nullptr, nullptr}, ev_scan[2] = {nullptr, nullptr},
             ev_resc[2] = {nullptr, nullptr};
  bool scan_recorded[2] = {false, false}, resc_recorded[2] = {false, false};
  int parity = 0;
  int *status = nullptr;   // sticky rescoring flags of the batches in flight
};
Pipeline &pipeline();
int pipeline_init();
int pipeline_drain();   // waits for A and B; reports the sticky status of the drained batches

// Stage timers (HIP events on the library's stream).
struct ProfScope {
  int slot = -1;
  explicit ProfScope(const char *stage);
  ~ProfScope();
};
void prof_add_scanned(int64_t vectors);
unsigned long long *prof_scanned_dev();   // device accumulator (nullptr on allocation failure)
bool prof_enabled();
bool prof_counts();

// Device copy of a packed spectra set (pointers are device pointers).
struct DevPeaks {
  int32_t n = 0;
  int64_t n_peaks = 0;
  const int32_t *offsets = nullptr;
  const float *mz = nullptr;
  const float *intensity = nullptr;
  const uint8_t *charge = nullptr;  // may be nullptr
  const double *precursor_mz = nullptr;
  const int32_t *precursor_charge = nullptr;
  // library only (asl_library): every spectrum's peaks as ONE record, [mz f32 x n][charge u8 x n]
  // [intensity f32 x n, at float index rec_int0(n)] at a 16-byte aligned offset (RowMeta::rec4).
  // The rescoring stream reads m/z and charges -- the first 5 n bytes, one or two cache lines --
  // and the intensities only of the peaks that match
  const uint8_t *records = nullptr;
};
// float index of the first intensity in a record of n peaks; the record's size in bytes
__host__ __device__ inline int rec_int0(int n) { return n + ((n + 3) >> 2); }
__host__ __device__ inline uint64_t rec_bytes(uint64_t n) { return 4ull * (uint64_t)rec_int0((int)n) + 4ull * n; }

// spectral_library.py:421-427 (numexpr evaluates in float64): is library precursor m/z `lib` (the
// float32 column of spec_info) inside the query's window?
__device__ __forceinline__ bool precursor_ok(double q, float lib, int charge, double tol,
                                             int mode) {
  const double l = (double)lib;
  if (mode == ASL_TOL_DA) return fabs(q - l) * (double)charge <= tol;
  return fabs(q - l) / l * 1000000.0 <= tol;
}

// The precursor-window post-filter (spectral_library.py:441-446: AFTER the top-k) applied where the
// top-k ends: a scan's set-mode finish that is given one writes only the hits that pass, compacted at
// the front of the row, and their number -- the rescoring then walks ~1/6 of a 1 024-wide row and
// gathers no window column (33.5 M random 4-byte gathers per batch: 0.27 ms,
// profiles/r06_rescore_prefilter.txt). idpay[slot] = (vector id, bits of the library's float32
// window column for that id; NaN = never a candidate): the value arrives with the id the finish
// gathers anyway. count[q] = -1: the row holds the k UNFILTERED hits (the exact-flush fallback of a
// row with mass ties emits through another path): the rescoring filters that row itself.
struct ScanPostFilter {
  const int2 *idpay = nullptr;      // nullptr: no filter (every other field unused)
  const double *q_pmz = nullptr;    // per query
  int32_t *count = nullptr;         // per query, out
  double tol = 0.0;
  int mode = ASL_TOL_DA;
  int charge = 0;
};

// What search.hip hands an index for its NEXT search (index_set_post_filter): the library's window
// column by vector id (device, float32, NaN = never a candidate; the index keeps (id, value) pairs
// per storage slot, rebuilt when the column or the lists change), the queries' precursor m/z and the
// window; count[nq] receives the rows' lengths. index_post_filter_applied() says whether the scan
// that ran could take it (set-mode int32 rows of the layout-specific scans, k <= 1280); if not, the
// rows are the k unfiltered hits as ever.
struct IndexPostFilter {
  const float *payload = nullptr;
  int64_t n = 0;
  const double *q_pmz = nullptr;
  int32_t *count = nullptr;
  double tol = 0.0;
  int mode = ASL_TOL_DA;
  int charge = 0;
};

struct PeaksStage {  // stages an asl_peaks_t whose arrays may be on the host
  In<int32_t> offsets, pcharge;
  In<float> mz, intensity;
  In<uint8_t> charge;
  In<double> pmz;
  DevPeaks dev;
  int init(const asl_peaks_t *p);
};
bool peaks_on_device(const asl_peaks_t *p);

// Inclusive prefix sum over the 64 lanes of a wave in seven DPP adds (row_shr 1 / 2 / 3, row_shr 4
// and 8 under bank masks, row_bcast 15 and 31 under row masks) -- no LDS crossbar round trips:
// the ds_bpermute chain this replaces (six dependent __shfl_up) was ~0.3 us of latency per scan,
// and a top-k finish runs three of them. All 64 lanes must be active.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t scan_dpp(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  uint32_t t = x + scan_dpp<0x111, 0xf, 0xf>(x);
  t += scan_dpp<0x112, 0xf, 0xf>(x);
  t += scan_dpp<0x113, 0xf, 0xf>(x);
  t += scan_dpp<0x114, 0xf, 0xe>(t);
  t += scan_dpp<0x118, 0xf, 0xc>(t);
  t += scan_dpp<0x142, 0xa, 0xf>(t);
  t += scan_dpp<0x143, 0xc, 0xf>(t);
  return t;
}
// Maximum over the 64 lanes of a wave, valid in lane 63 (same ladder with max for +).
__device__ __forceinline__ int wave_max_to_lane63(int x) {
  auto mx = [](int a, uint32_t b) { return a > (int)b ? a : (int)b; };   // values >= 0: the DPP fill 0 is neutral
  int t = mx(x, scan_dpp<0x111, 0xf, 0xf>((uint32_t)x));
  t = mx(t, scan_dpp<0x112, 0xf, 0xf>((uint32_t)x));
  t = mx(t, scan_dpp<0x113, 0xf, 0xf>((uint32_t)x));
  t = mx(t, scan_dpp<0x114, 0xf, 0xe>((uint32_t)t));
  t = mx(t, scan_dpp<0x118, 0xf, 0xc>((uint32_t)t));
  t = mx(t, scan_dpp<0x142, 0xa, 0xf>((uint32_t)t));
  t = mx(t, scan_dpp<0x143, 0xc, 0xf>((uint32_t)t));
  return t;
}

}  // namespace asl

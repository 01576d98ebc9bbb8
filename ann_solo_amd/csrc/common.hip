// common.hip -- error state, stream, pointer classification, stage timers.
#include <cstdlib>

#include "common.hpp"

#include <mutex>

namespace asl {

static thread_local std::string g_err;
static hipStream_t g_stream = nullptr;

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
void clear_error() { g_err.clear(); }

hipStream_t stream() { return g_stream; }

static Pipeline g_pipe;
static int ensure_device_raw();
int ensure_device() {
  ASL_TRY(ensure_device_raw());
  // batches of the two-stream pipeline still in flight use the handles' scratch buffers: any
  // other entry point first waits for them (and reports their deferred status)
  if (g_pipe.inflight && !g_pipe.in_call) ASL_TRY(pipeline_drain());
  return ASL_OK;
}
static int ensure_device_raw() {
  static int state = 0;  // 0 unknown, 1 ok, -1 none
  if (state == 0) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    state = (e == hipSuccess && n > 0) ? 1 : -1;
    if (state < 0) (void)hipGetLastError();
  }
  if (state < 0)
    return fail(ASL_ERR_NO_DEVICE,
                "no HIP device available: libannsolo_mi has no CPU fallback");
  return ASL_OK;
}

bool is_device_ptr(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain host memory is reported as an error
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int sync_stream() {
  HIP_TRY(hipStreamSynchronize(g_stream));
  return ASL_OK;
}

StreamScope::StreamScope(hipStream_t s) : prev(g_stream) { g_stream = s; }
StreamScope::~StreamScope() { g_stream = prev; }

Pipeline &pipeline() { return g_pipe; }

int pipeline_init() {
  Pipeline &p = g_pipe;
  if (p.A) return ASL_OK;
  // non-blocking: no implicit ordering against the null stream (the caller's, PyTorch's) --
  // the ordering that matters is expressed with events
  // Priorities. Every phase of a step keeps the chip busy, so overlapping conserves the work: the
  // step is the sum of the kernels' stand-alone times whichever stream goes first; the priorities
  // only decide WHICH kernel the short front stage (A: encode + coarse quantiser, ~0.9 ms of a
  // 14 ms step) lands on. Until round 6 the front had the lowest priority and filled what scan +
  // rescoring left (profiles/r02_pipeline_ab.txt; round 5 re-measured all three arrangements: no
  // difference in the step). Since the precursor filter moved into the scan's finish the rescoring
  // is shorter, and the front at the HIGHEST priority -- done early inside the scan, which then
  // runs undisturbed -- measures 0.9 % (32 768-query batches) to 2 % (16 384) faster
  // (profiles/r06_rescore_prefilter.txt).
  int least = 0, greatest = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
  HIP_TRY(hipStreamCreateWithPriority(&p.A, hipStreamNonBlocking, greatest));
  HIP_TRY(hipStreamCreateWithPriority(&p.B, hipStreamNonBlocking, least));
  HIP_TRY(hipStreamCreateWithPriority(&p.C, hipStreamNonBlocking, least));
  HIP_TRY(hipEventCreateWithFlags(&p.ev_in, hipEventDisableTiming));
  for (int i = 0; i < 2; i++) {
    HIP_TRY(hipEventCreateWithFlags(&p.ev_front[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_scan[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_resc[i], hipEventDisableTiming));
  }
  HIP_TRY(hipMalloc((void **)&p.status, sizeof(int)));
  HIP_TRY(hipMemset(p.status, 0, sizeof(int)));
  return ASL_OK;
}

int rescore_status_error(int status_bits);

int pipeline_drain() {
  Pipeline &p = g_pipe;
  if (!p.inflight) return ASL_OK;
  p.inflight = false;
  HIP_TRY(hipStreamSynchronize(p.A));
  HIP_TRY(hipStreamSynchronize(p.B));
  HIP_TRY(hipStreamSynchronize(p.C));
  int st = 0;
  HIP_TRY(hipMemcpy(&st, p.status, sizeof(int), hipMemcpyDeviceToHost));
  if (st) HIP_TRY(hipMemset(p.status, 0, sizeof(int)));
  return rescore_status_error(st);
}

// ---------------------------------------------------------------- profiling
struct StageProf {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  double total_ms = 0;
  int64_t launches = 0;
};
static std::map<std::string, StageProf> g_prof;
static int g_prof_on = 0;   // 0 off, 1 every stage, 2 the list scan only (two events per step)
static int64_t g_scanned = 0;
static std::vector<std::pair<StageProf *, std::pair<hipEvent_t, hipEvent_t>>> g_open;

ProfScope::ProfScope(const char *stage) {
  if (!g_prof_on || (g_prof_on == 2 && std::strcmp(stage, "scan") != 0)) return;
  StageProf &sp = g_prof[stage];
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!sp.pool.empty()) {
    ev = sp.pool.back();
    sp.pool.pop_back();
  } else {
    if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess)
      return;
  }
  (void)hipEventRecord(ev.first, g_stream);
  slot = (int)g_open.size();
  g_open.push_back({&sp, ev});
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  auto o = g_open[slot];
  (void)hipEventRecord(o.second.second, g_stream);
  o.first->pending.push_back(o.second);
  o.first->launches++;
  if (slot == (int)g_open.size() - 1) g_open.pop_back();
}
bool prof_enabled() { return g_prof_on != 0; }
bool prof_counts() { return g_prof_on == 1; }   // the scanned-vector count costs a kernel per scan: not at level 2
void prof_add_scanned(int64_t v) {
  if (g_prof_on == 1) g_scanned += v;
}
// device-side accumulator of scanned vectors: kernels add to it, nothing waits inside a step
static unsigned long long *g_scanned_dev = nullptr;
unsigned long long *prof_scanned_dev() {
  if (!g_scanned_dev) {
    if (hipMalloc((void **)&g_scanned_dev, sizeof(unsigned long long)) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(g_scanned_dev, 0, sizeof(unsigned long long), g_stream);
  }
  return g_scanned_dev;
}

static void prof_collect(StageProf &sp) {
  for (auto &ev : sp.pending) {
    float ms = 0;
    if (hipEventSynchronize(ev.second) == hipSuccess &&
        hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess)
      sp.total_ms += ms;
    sp.pool.push_back(ev);
  }
  sp.pending.clear();
}

}  // namespace asl

using namespace asl;

extern "C" {

const char *asl_last_error(void) { return g_err.c_str(); }
const char *asl_version(void) { return "annsolo_mi 0.1.0 (gfx950)"; }

int asl_get_num_gpus(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int asl_set_device(int device) {
  ASL_TRY(ensure_device());
  HIP_TRY(hipSetDevice(device));
  return ASL_OK;
}

int asl_set_stream(void *s) {
  g_stream = (hipStream_t)s;
  return ASL_OK;
}

int asl_synchronize(void) {
  clear_error();
  ASL_TRY(ensure_device());   // drains the pipeline (deferred status of its batches)
  return sync_stream();
}

int asl_set_pipeline(int on) {
  clear_error();
  ASL_TRY(ensure_device());
  if (on < 0 || on > 3) return fail(ASL_ERR_INVALID, "set_pipeline: 0 (off), 1 / 2 (two streams) or 3");
  if (on) ASL_TRY(pipeline_init());
  g_pipe.on = on != 0;
  g_pipe.streams = on == 3 ? 3 : 2;
  return ASL_OK;
}

int asl_profile_enable(int on) {
  g_prof_on = on == 2 ? 2 : (on != 0);
  return ASL_OK;
}

int asl_profile_reset(void) {
  for (auto &kv : g_prof) {
    prof_collect(kv.second);
    kv.second.total_ms = 0;
    kv.second.launches = 0;
  }
  g_scanned = 0;
  if (g_scanned_dev) (void)hipMemsetAsync(g_scanned_dev, 0, sizeof(unsigned long long), g_stream);
  return ASL_OK;
}

int asl_profile_get(const char *stage, double *total_ms, int64_t *launches) {
  auto it = g_prof.find(stage ? stage : "");
  if (it == g_prof.end()) {
    if (total_ms) *total_ms = 0;
    if (launches) *launches = 0;
    return ASL_OK;
  }
  prof_collect(it->second);
  if (total_ms) *total_ms = it->second.total_ms;
  if (launches) *launches = it->second.launches;
  return ASL_OK;
}

int64_t asl_profile_scanned_vectors(void) {
  unsigned long long dev = 0;
  if (g_scanned_dev && hipMemcpyAsync(&dev, g_scanned_dev, sizeof(dev), hipMemcpyDeviceToHost, g_stream) == hipSuccess)
    (void)hipStreamSynchronize(g_stream);
  return g_scanned + (int64_t)dev;
}

}  // extern "C"

namespace asl {

int PeaksStage::init(const asl_peaks_t *p) {
  if (!p || p->n < 0) return fail(ASL_ERR_INVALID, "peaks: null or negative n");
  dev = DevPeaks();
  dev.n = p->n;
  if (p->n == 0) return ASL_OK;
  if (!p->offsets || !p->precursor_mz || !p->precursor_charge)
    return fail(ASL_ERR_INVALID, "peaks: offsets/precursor arrays are required");
  int32_t last = 0;
  if (p->n_peaks > 0) {          // the caller knows offsets[n]: no read-back, no synchronisation
    if (p->n_peaks > 0x7fffffffLL) return fail(ASL_ERR_INVALID, "peaks: more than 2^31-1 peaks");
    last = (int32_t)p->n_peaks;
  } else if (is_device_ptr(p->offsets)) {
    HIP_TRY(hipMemcpyAsync(&last, p->offsets + p->n, sizeof(int32_t), hipMemcpyDeviceToHost,
                           stream()));
    ASL_TRY(sync_stream());
  } else {
    last = p->offsets[p->n];
  }
  if (last < 0) return fail(ASL_ERR_INVALID, "peaks: negative offsets");
  dev.n_peaks = last;
  if (last > 0 && (!p->mz || !p->intensity))
    return fail(ASL_ERR_INVALID, "peaks: mz/intensity arrays are required");
  ASL_TRY(offsets.init(p->offsets, (size_t)p->n + 1));
  ASL_TRY(mz.init(p->mz, (size_t)last));
  ASL_TRY(intensity.init(p->intensity, (size_t)last));
  ASL_TRY(charge.init(p->charge, (size_t)last));
  ASL_TRY(pmz.init(p->precursor_mz, (size_t)p->n));
  ASL_TRY(pcharge.init(p->precursor_charge, (size_t)p->n));
  dev.offsets = offsets.d;
  dev.mz = mz.d;
  dev.intensity = intensity.d;
  dev.charge = charge.d;
  dev.precursor_mz = pmz.d;
  dev.precursor_charge = pcharge.d;
  return ASL_OK;
}

bool peaks_on_device(const asl_peaks_t *p) {
  return p && is_device_ptr(p->offsets) && is_device_ptr(p->mz) && is_device_ptr(p->intensity) &&
         is_device_ptr(p->precursor_mz) && is_device_ptr(p->precursor_charge) &&
         (!p->charge || is_device_ptr(p->charge));
}

}  // namespace asl

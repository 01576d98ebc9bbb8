// process.hip -- batched peak preprocessing (replaces process_spectrum,
// /root/reference/src/ann_solo/spectrum.py:57-119, i.e. the spectrum_utils 0.3.x calls
// set_mz_range / round / remove_precursor_peak / filter_intensity / scale_intensity + L2 norm and the
// validity checks of spectrum.py:13-36). PARITY UNPINNED: spectrum_utils is an
// un-vendored dependency; the kernel is bit-identical to oracle/orc_process_spectrum.
//
// One workgroup per raw spectrum (<= 4096 peaks, ascending m/z). The intensity ranking is a
// register-blocked bitonic sort of (intensity, index) keys sized to the spectrum
// (256 / 1024 / 4096); rank r < max_peaks with intensity above min_intensity * base peak
// survives; survivors are compacted in m/z order by a prefix scan; the L2 norm is the
// canonical ascending fmaf chain.
#include "common.hpp"
#include "hist_topk.hpp"

namespace asl {

constexpr int PS_NT = 256, PS_MAXN = 4096;

struct ProcParams {
  double min_mz, max_mz, rp_tol, min_intensity, min_mz_range;
  int remove_precursor, max_peaks, scaling, min_peaks, resolution;
};

// MsmsSpectrum.round's m/z rounding (numba np.round_ on a float: evaluated in double, ties to
// even, stored as float32) -- oracle: orc_round_mz
__device__ __forceinline__ float round_mz(float mz, int decimals) {
  double p = 1.0;
  const int a = decimals < 0 ? -decimals : decimals;
  for (int i = 0; i < a; ++i) p *= 10.0;
  const double x = (double)mz;
  return (float)(decimals >= 0 ? rint(x * p) / p : rint(x / p) * p);
}

__device__ __forceinline__ bool wg_valid(const float *mz, const uint8_t *keep, int n, int tid,
                                         int *s3 /* cnt, first, last */, int min_peaks,
                                         double min_range) {
  if (tid == 0) {
    s3[0] = 0;
    s3[1] = 0x7fffffff;
    s3[2] = -1;
  }
  __syncthreads();
  int c = 0, f = 0x7fffffff, l = -1;
  for (int i = tid; i < n; i += PS_NT)
    if (keep[i]) {
      ++c;
      f = min(f, i);
      l = max(l, i);
    }
  if (c) {
    atomicAdd(&s3[0], c);
    atomicMin(&s3[1], f);
    atomicMax(&s3[2], l);
  }
  __syncthreads();
  const bool ok = s3[0] >= min_peaks && s3[0] > 0 &&
                  (double)(mz[s3[2] >= 0 ? s3[2] : 0] - mz[s3[1] < n ? s3[1] : 0]) >= min_range;
  __syncthreads();
  return ok;
}

// ROUND: config.resolution is set -- the spectrum is staged in LDS, its m/z rounded and peaks
// with equal rounded m/z merged (spectrum.py:84-85) before the remaining steps run on the copy.
template <bool ROUND>
__global__ __launch_bounds__(PS_NT) void process_kernel(
    DevPeaks raw, ProcParams P, float *__restrict__ out_mz, float *__restrict__ out_int,
    int32_t *__restrict__ out_src, int32_t *__restrict__ out_count,
    uint8_t *__restrict__ out_valid, int *status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64 *keys = reinterpret_cast<u64 *>(smem);                       // [PS_MAXN]
  float *val = reinterpret_cast<float *>(keys + PS_MAXN);          // [PS_MAXN]
  uint8_t *keep = reinterpret_cast<uint8_t *>(val + PS_MAXN);      // [PS_MAXN]
  int *ctl = reinterpret_cast<int *>(keep + PS_MAXN);              // 16 ints
  float *w_mz = reinterpret_cast<float *>(ctl + 16);               // ROUND only: [PS_MAXN] x 3
  float *w_int = w_mz + PS_MAXN;
  int *w_src = reinterpret_cast<int *>(w_int + PS_MAXN);
  const int s = blockIdx.x, tid = threadIdx.x;
  const int o = raw.offsets[s];
  int n = raw.offsets[s + 1] - o;
  const float *mz = raw.mz + o, *inten = raw.intensity + o;
  auto bail = [&]() {
    if (tid == 0) {
      out_count[s] = 0;
      out_valid[s] = 0;
    }
  };
  if (n > PS_MAXN) {
    if (tid == 0) atomicOr(status, 1);
    bail();
    return;
  }
  if (n <= 0) {
    bail();
    return;
  }
  // spectrum.py:79 set_mz_range (inclusive) [+ :90-92 remove_precursor_peak(tol, 'Da', 2)]
  const double pmz = raw.precursor_mz[s];
  const int pz = raw.precursor_charge[s];
  for (int i = tid; i < n; i += PS_NT) {
    const double m = (double)mz[i];
    keep[i] = m >= P.min_mz && m <= P.max_mz;
  }
  __syncthreads();
  if (!wg_valid(mz, keep, n, tid, ctl, P.min_peaks, P.min_mz_range)) {
    bail();
    return;
  }
  if (ROUND) {
    int *head = reinterpret_cast<int *>(val);   // scratch: val is not live yet
    for (int i = tid; i < n; i += PS_NT) {
      w_mz[i] = keep[i] ? round_mz(mz[i], P.resolution) : mz[i];
      w_int[i] = inten[i];
      w_src[i] = i;
    }
    __syncthreads();
    for (int i = tid; i < n; i += PS_NT)
      head[i] = keep[i] && (i == 0 || !keep[i - 1] || w_mz[i - 1] != w_mz[i]);
    __syncthreads();
    for (int i = tid; i < n; i += PS_NT) {
      if (!head[i]) continue;
      int best = i;
      float sum = w_int[i], bv = w_int[i];
      for (int j = i + 1; j < n && keep[j] && w_mz[j] == w_mz[i]; ++j) {
        const float v = w_int[j];
        sum += v;
        if (v > bv) {
          bv = v;
          best = j;
        }
      }
      w_int[i] = sum;    // only this head reads its group's entries
      w_src[i] = best;
    }
    __syncthreads();
    for (int i = tid; i < n; i += PS_NT) keep[i] = (uint8_t)head[i];
    __syncthreads();
    mz = w_mz;
    inten = w_int;
    if (!wg_valid(mz, keep, n, tid, ctl, P.min_peaks, P.min_mz_range)) {
      bail();
      return;
    }
  }
  if (P.remove_precursor) {
    const double adduct = 1.0072766;
    const double neutral = (pmz - adduct) * (double)pz;
    for (int i = tid; i < n; i += PS_NT) {
      if (!keep[i]) continue;
      const double m = (double)mz[i];
      bool k = true;
      for (int charge = pz; charge >= 1; --charge)
        for (int iso = 0; iso <= 2; ++iso) {
          const double rm = (neutral + iso) / charge + adduct;
          if (fabs(m - rm) <= P.rp_tol) k = false;
        }
      keep[i] = k;
    }
    __syncthreads();
    if (!wg_valid(mz, keep, n, tid, ctl, P.min_peaks, P.min_mz_range)) {
      bail();
      return;
    }
  }
  // spectrum.py:97-99 filter_intensity: rank by (intensity desc, index desc)
  const int nsort = n <= 256 ? 256 : (n <= 1024 ? 1024 : 4096);
  for (int i = tid; i < nsort; i += PS_NT)
    keys[i] = (i < n && keep[i]) ? (((u64)f2ord(inten[i]) << 32) | (u64)(uint32_t)i) : 0ull;
  __syncthreads();
  const int lead = P.max_peaks < nsort ? P.max_peaks : nsort;
  if (nsort == 256)
    block_sort_desc<PS_NT, 1>(keys, tid, lead);
  else if (nsort == 1024)
    block_sort_desc<PS_NT, 4>(keys, tid, lead);
  else
    block_sort_desc<PS_NT, 16>(keys, tid, lead);
  for (int i = tid; i < n; i += PS_NT) keep[i] = 0;
  __syncthreads();
  {
    const double thresh = P.min_intensity * (double)ord2f((uint32_t)(keys[0] >> 32));
    if (tid < lead) {
      const u64 key = keys[tid];
      if (key != 0ull) {
        const float v = ord2f((uint32_t)(key >> 32));
        const int idx = (int)(uint32_t)key;
        if ((double)v > thresh) {
          keep[idx] = 1;
          val[idx] = P.scaling == 1 ? (float)(P.max_peaks - tid)
                                    : (P.scaling == 2 ? __builtin_sqrtf(v) : v);
        }
      }
    }
  }
  __syncthreads();
  if (!wg_valid(mz, keep, n, tid, ctl, P.min_peaks, P.min_mz_range)) {
    bail();
    return;
  }
  // compaction in m/z order
  constexpr int PER = PS_MAXN / PS_NT;
  int cnt = 0;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid * PER + u;
    cnt += (i < n && keep[i]) ? 1 : 0;
  }
  int tot;
  int pos = block_excl_scan256(cnt, ctl + 4, tid, tot);
  float *c_val = reinterpret_cast<float *>(keys);  // compacted scaled values (sort buffer is free)
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid * PER + u;
    if (i < n && keep[i]) {
      out_mz[(size_t)s * P.max_peaks + pos] = mz[i];
      out_src[(size_t)s * P.max_peaks + pos] = ROUND ? w_src[i] : i;
      c_val[pos] = val[i];
      ++pos;
    }
  }
  __syncthreads();
  // spectrum.py:112 L2 norm: ascending fmaf chain, sqrtf, IEEE divide
  if (tid == 0) {
    float acc = 0.0f;
    for (int t = 0; t < tot; ++t) acc = __builtin_fmaf(c_val[t], c_val[t], acc);
    reinterpret_cast<float *>(ctl)[8] = __builtin_sqrtf(acc);
  }
  __syncthreads();
  const float nrm = reinterpret_cast<float *>(ctl)[8];
  for (int t = tid; t < tot; t += PS_NT) out_int[(size_t)s * P.max_peaks + t] = c_val[t] / nrm;
  if (tid == 0) {
    out_count[s] = tot;
    out_valid[s] = 1;
  }
}

}  // namespace asl

using namespace asl;

extern "C" int asl_process_batch(const asl_peaks_t *raw, const asl_process_params_t *p,
                                 float *out_mz, float *out_intensity, int32_t *out_src,
                                 int32_t *out_count, uint8_t *out_valid) {
  clear_error();
  if (!raw || !p || !out_mz || !out_intensity || !out_count || !out_valid)
    return fail(ASL_ERR_INVALID, "process_batch: null argument");
  if (p->max_peaks <= 0 || p->max_peaks > PS_NT)
    return fail(ASL_ERR_INVALID, "process_batch: max_peaks must be in 1..%d", PS_NT);
  const int n = raw->n;
  if (n == 0) return ASL_OK;
  ASL_TRY(ensure_device());
  PeaksStage R;
  ASL_TRY(R.init(raw));
  Out<float> o_mz, o_int;
  Out<int32_t> o_src, o_cnt;
  Out<uint8_t> o_val;
  const size_t slots = (size_t)n * p->max_peaks;
  ASL_TRY(o_mz.init(out_mz, slots));
  ASL_TRY(o_int.init(out_intensity, slots));
  DevBuf<int32_t> src_tmp;
  if (out_src) {
    ASL_TRY(o_src.init(out_src, slots));
  } else {
    ASL_TRY(src_tmp.reserve(slots));
    o_src.d = src_tmp.p;
  }
  ASL_TRY(o_cnt.init(out_count, n));
  ASL_TRY(o_val.init(out_valid, n));
  DevBuf<int> status;
  ASL_TRY(status.reserve(1));
  HIP_TRY(hipMemsetAsync(status.p, 0, sizeof(int), stream()));
  if (p->round_mz && (p->resolution < -6 || p->resolution > 12))
    return fail(ASL_ERR_INVALID, "process_batch: resolution must be in -6..12 decimals");
  ProcParams P{p->min_mz, p->max_mz, p->remove_precursor_tolerance, p->min_intensity,
               p->min_mz_range, p->remove_precursor, p->max_peaks, p->scaling, p->min_peaks,
               p->resolution};
  const bool round = p->round_mz != 0;
  const size_t lds = (size_t)PS_MAXN * (8 + 4 + 1) + 64 + (round ? (size_t)PS_MAXN * 12 : 0);
  const void *fn = round ? (const void *)process_kernel<true> : (const void *)process_kernel<false>;
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  {
    ProfScope ps("process");
    if (round)
      hipLaunchKernelGGL(process_kernel<true>, dim3(n), dim3(PS_NT), lds, stream(), R.dev, P,
                         o_mz.d, o_int.d, o_src.d, o_cnt.d, o_val.d, status.p);
    else
      hipLaunchKernelGGL(process_kernel<false>, dim3(n), dim3(PS_NT), lds, stream(), R.dev, P,
                         o_mz.d, o_int.d, o_src.d, o_cnt.d, o_val.d, status.p);
    ASL_CHECK_LAUNCH();
  }
  ASL_TRY(o_mz.finish());
  ASL_TRY(o_int.finish());
  ASL_TRY(o_src.finish());
  ASL_TRY(o_cnt.finish());
  ASL_TRY(o_val.finish());
  int st = 0;
  HIP_TRY(hipMemcpyAsync(&st, status.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
  ASL_TRY(sync_stream());
  if (st) return fail(ASL_ERR_CAPACITY, "process_batch: a spectrum has more than %d peaks", PS_MAXN);
  return ASL_OK;
}

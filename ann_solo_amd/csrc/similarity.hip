// similarity.hip -- batched spectrum-spectrum-match similarity features.
// Replaces the per-SSM Python of SpectrumSimilarityCalculator
// (/root/reference/src/ann_solo/spectrum_similarity.py:13-730) as it is called by
// _compute_ssm_features (/root/reference/src/ann_solo/utils.py:344-456): the 33 similarity
// columns (full spectrum and `top` most intense library peaks) of every query's best match.
//
// One 64-lane wavefront per SSM, both spectra and every derived list in LDS. The O(n^2)
// parts (top-5 membership, Kendall pair counts, average ranks) and the reductions are
// lane-parallel, the exact Kendall p-value recurrence (<= 33 rows of <= 265 counts) too since
// round 5 (five counts per lane, a wave scan per row). Arithmetic is fp64 on the fp32 peaks; the reference sums float32
// arrays, so agreement with it is ~1e-6 (tolerance 1e-5), with the oracle ~1e-12.
#include "common.hpp"

namespace asl {

constexpr int SIM_MAXP = 256;    // peaks per spectrum (largest instantiation)
constexpr int SIM_WAVES = 2;     // SSMs per workgroup
constexpr int SIM_KC = 272;      // exact Kendall recurrence: c <= 33*32/4 = 264

// MAXP = 128 halves the footprint (15 KB per wave: 10 waves per CU instead of 4); spectra are
// <= 50 peaks with the reference's defaults, the 256-peak instantiation is the fallback.
template <int MAXP>
struct SimLdsT {
  float q_mz[MAXP], q_int[MAXP], l_mz[MAXP], l_int[MAXP];
  float mq[MAXP], ml[MAXP], mzq[MAXP], mzl[MAXP];   // matched, match order
  float tq[MAXP], tl[MAXP], tzq[MAXP], tzl[MAXP];   // matched & in top
  float uq[MAXP], ul[MAXP], tul[MAXP];              // unmatched lists
  double x[2 * MAXP], y[MAXP], rx[MAXP], ry[MAXP];  // x also holds the merged spectrum
  double kc[2][SIM_KC];
  uint8_t used_q[MAXP], used_l[MAXP], in_top[MAXP];
};

__device__ __forceinline__ void sim_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ long long wsum_ll(long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ bool wall(bool p) { return __ballot(!p) == 0ull; }

// order-preserving append of the flagged items of positions [0, n) to dst lists
template <class F>
__device__ __forceinline__ int compact(int lane, int n, F &&emit_if) {
  int total = 0;
  for (int b = 0; b < n; b += 64) {
    const int i = b + lane;
    const bool f = i < n && emit_if(i, -1);
    const unsigned long long bal = __ballot(f);
    if (f) emit_if(i, total + __popcll(bal & ((1ull << lane) - 1ull)));
    total += __popcll(bal);
  }
  return total;
}

__device__ __forceinline__ double lcomb(double n, double k) {
  if (k < 0.0 || k > n) return -INFINITY;
  return lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0);
}

// scipy.stats.entropy of x[0..n) (natural log)
__device__ double entropy_of(int lane, const double *x, int n) {
  double s = 0.0;
  for (int i = lane; i < n; i += 64) s += x[i];
  s = wsum(s);
  double h = 0.0;
  for (int i = lane; i < n; i += 64) {
    const double p = x[i] / s;
    if (p > 0.0) h -= p * log(p);
  }
  return wsum(h);
}

// spectrum_similarity.py:703-730; x is overwritten by its weighted version
__device__ double spectrum_entropy(int lane, double *x, int n, bool weighted) {
  const double h = entropy_of(lane, x, n);
  if (!weighted || h > 3.0) return h;
  const double w = 0.25 + (1.0 - 0.25) / 3.0 * h;
  for (int i = lane; i < n; i += 64) x[i] = pow(x[i], w);
  sim_sync();
  return entropy_of(lane, x, n);
}

__device__ void avg_ranks(int lane, const double *x, int n, double *r) {
  for (int i = lane; i < n; i += 64) {
    int less = 0, eq = 0;
    const double xi = x[i];
    for (int j = 0; j < n; j++) {
      less += x[j] < xi;
      eq += x[j] == xi;
    }
    r[i] = less + 0.5 * (eq + 1);
  }
  sim_sync();
}

// scipy.stats.pearsonr statistic (NaN -> 0 as the reference maps it)
__device__ double pearson(int lane, const double *x, const double *y, int n) {
  if (n < 2) return 0.0;
  bool cx = true, cy = true;
  double mx = 0, my = 0;
  const double x0 = x[0], y0 = y[0];
  for (int i = lane; i < n; i += 64) {
    cx = cx && x[i] == x0;
    cy = cy && y[i] == y0;
    mx += x[i];
    my += y[i];
  }
  if (wall(cx) || wall(cy)) return 0.0;
  mx = wsum(mx) / n;
  my = wsum(my) / n;
  double xmax = 0, ymax = 0;
  for (int i = lane; i < n; i += 64) {
    xmax = fmax(xmax, fabs(x[i] - mx));
    ymax = fmax(ymax, fabs(y[i] - my));
  }
  xmax = wmax(xmax);
  ymax = wmax(ymax);
  double sx = 0, sy = 0;
  for (int i = lane; i < n; i += 64) {
    const double a = (x[i] - mx) / xmax, b = (y[i] - my) / ymax;
    sx += a * a;
    sy += b * b;
  }
  const double nx = xmax * sqrt(wsum(sx)), ny = ymax * sqrt(wsum(sy));
  double r = 0;
  for (int i = lane; i < n; i += 64) r += (x[i] - mx) / nx * (y[i] - my) / ny;
  r = fmax(-1.0, fmin(1.0, wsum(r)));
  if (n == 2) r = rint(r);
  return isnan(r) ? 0.0 : r;
}

// -log(p) of scipy.stats.kendalltau (tau-b, method 'auto', two-sided) of (x, y)[0..n)
__device__ double kendall_neglogp(int lane, const float *x, const float *y, int n, double *kc0,
                                  double *kc1) {
  if (n < 2) return 0.0;
  long long dis = 0, xtie = 0, ytie = 0, ntie = 0;
  double tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0;   // tie-group statistics (asymptotic variance)
  for (int i = lane; i < n; i += 64) {
    const float xi = x[i], yi = y[i];
    int cxe = 0, cye = 0;
    bool firstx = true, firsty = true;
    for (int j = 0; j < n; j++) {
      const bool ex = x[j] == xi, ey = y[j] == yi;
      cxe += ex;
      cye += ey;
      if (j < i) {
        firstx = firstx && !ex;
        firsty = firsty && !ey;
      } else if (j > i) {
        xtie += ex;
        ytie += ey;
        ntie += ex && ey;
        if (!ex && !ey && ((xi < x[j]) != (yi < y[j]))) dis++;
      }
    }
    if (firstx && cxe > 1) {
      tx0 += (double)cxe * (cxe - 1.0) * (cxe - 2.0);
      tx1 += (double)cxe * (cxe - 1.0) * (2.0 * cxe + 5.0);
    }
    if (firsty && cye > 1) {
      ty0 += (double)cye * (cye - 1.0) * (cye - 2.0);
      ty1 += (double)cye * (cye - 1.0) * (2.0 * cye + 5.0);
    }
  }
  dis = wsum_ll(dis);
  xtie = wsum_ll(xtie);
  ytie = wsum_ll(ytie);
  ntie = wsum_ll(ntie);
  const long long tot = (long long)n * (n - 1) / 2;
  if (xtie == tot || ytie == tot) return 0.0;
  const long long con_minus_dis = tot - xtie - ytie + ntie - 2 * dis;
  const long long mn = dis < tot - dis ? dis : tot - dis;
  double p;
  if (xtie == 0 && ytie == 0 && (n <= 33 || mn <= 1)) {
    long long c = tot - dis;
    if (tot - c < c) c = tot - c;
    if (n == 2)
      p = 1.0;
    else if (c == 0)
      p = 2.0 / tgamma(n + 1.0);
    else if (c == 1)
      p = 2.0 / tgamma((double)n);
    else if (4 * c == (long long)n * (n - 1))
      p = 1.0;
    else {
      // counts of permutations of j items with <= i inversions, j = 3 .. n (Kendall's recurrence:
      // new[i] = P[i] - P[i - j], P = prefix sums of the old row), i <= c <= 264. All 64 lanes:
      // lane L holds entries 5 L .. 5 L + 4, a row step is a prefix sum inside the lane, a scan of
      // the lanes' totals and one exchange through LDS for P[i - j]. (On lane 0 alone -- 31 x 265
      // dependent LDS round trips per SSM -- this recurrence was most of the kernel's time.)
      constexpr int PER = 5;
      static_assert(PER * 64 >= SIM_KC, "");
      double v[PER];
#pragma unroll
      for (int u = 0; u < PER; u++) v[u] = lane * PER + u <= 1 ? 1.0 : 0.0;
      for (int j = 3; j <= n; j++) {
#pragma unroll
        for (int u = 1; u < PER; u++) v[u] += v[u - 1];
        double incl = v[PER - 1];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const double t = __shfl_up(incl, o, 64);
          if (lane >= o) incl += t;
        }
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0.0;
#pragma unroll
        for (int u = 0; u < PER; u++) v[u] += excl;
        if (j <= c) {          // wave-uniform
          sim_sync();
#pragma unroll
          for (int u = 0; u < PER; u++)
            if (lane * PER + u <= (int)c) kc0[lane * PER + u] = v[u];
          sim_sync();
#pragma unroll
          for (int u = 0; u < PER; u++) {
            const int i = lane * PER + u;
            if (i >= j && i <= (int)c) v[u] -= kc0[i - j];
          }
        }
      }
      double s = 0.0;
#pragma unroll
      for (int u = 0; u < PER; u++)
        if (lane * PER + u <= (int)c) s += v[u];
      s = wsum(s);
      p = 2.0 * s / tgamma(n + 1.0);
    }
    p = fmax(0.0, fmin(1.0, p));
  } else {
    const double x0 = wsum(tx0), x1 = wsum(tx1), y0 = wsum(ty0), y1 = wsum(ty1);
    const double m = (double)n * (n - 1.0);
    const double var = (m * (2.0 * n + 5.0) - x1 - y1) / 18.0 + (2.0 * xtie * ytie) / m +
                       x0 * y0 / (9.0 * m * (n - 2.0));
    const double z = (double)con_minus_dis / sqrt(var);
    p = erfc(fabs(z) / sqrt(2.0));
  }
  if (isnan(p)) return 0.0;
  const double v = -log(p);
  return v == 0.0 ? 0.0 : v;
}

struct SimView {
  int n, n_ul;
  const float *mq, *ml, *mzq, *mzl, *ul;
};

__device__ double cosine_of(int lane, const SimView &v, bool renorm) {
  if (!v.n) return 0.0;
  double d = 0, a = 0, b = 0;
  for (int i = lane; i < v.n; i += 64) {
    const double q = v.mq[i], l = v.ml[i];
    d += q * l;
    a += q * q;
    b += l * l;
  }
  d = wsum(d);
  return renorm ? d / (sqrt(wsum(a)) * sqrt(wsum(b))) : d;
}

__device__ double mse_of(int lane, const float *a, const float *b, int n) {
  if (!n) return INFINITY;
  double s = 0;
  for (int i = lane; i < n; i += 64) {
    const double d = (double)(a[i] - b[i]);   // float32 subtraction, as in the reference
    s += d * d;
  }
  return wsum(s) / n;
}

__device__ double scribe_of(int lane, const SimView &v) {
  if (!v.n) return 0.0;
  double den = 0;
  for (int i = lane; i < v.n; i += 64) {
    const double d = (double)(v.mq[i] - v.ml[i]);
    den += d * d;
  }
  for (int i = lane; i < v.n_ul; i += 64) den += (double)v.ul[i] * (double)v.ul[i];
  den = wsum(den);
  return den == 0.0 ? 10.0 : log(1.0 / den);
}

struct SimScratch {
  double *x, *y, *rx, *ry;
};

__device__ double corr_of(int lane, const SimView &v, bool spearman, const SimScratch &S) {
  if (!v.n) return 0.0;
  const int n = v.n + v.n_ul;
  sim_sync();
  for (int i = lane; i < v.n; i += 64) {
    S.x[i] = v.mq[i];
    S.y[i] = v.ml[i];
  }
  for (int i = lane; i < v.n_ul; i += 64) {
    S.x[v.n + i] = 0.0;
    S.y[v.n + i] = v.ul[i];
  }
  sim_sync();
  if (!spearman) return pearson(lane, S.x, S.y, n);
  avg_ranks(lane, S.x, n, S.rx);
  avg_ranks(lane, S.y, n, S.ry);
  return pearson(lane, S.rx, S.ry, n);
}

template <int MAXP>
__global__ __launch_bounds__(64 * SIM_WAVES) void ssm_features_kernel(
    DevPeaks Qs, DevPeaks L, const int32_t *__restrict__ lib_rows,
    const uint32_t *__restrict__ pm_pairs, const int32_t *__restrict__ pm_count, int pm_stride,
    double n_bins, int top, double *__restrict__ out, int *status) {
  __shared__ SimLdsT<MAXP> W[SIM_WAVES];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * SIM_WAVES + wave;
  if (q >= Qs.n) return;
  SimLdsT<MAXP> &S = W[wave];
  const SimScratch scr{S.x, S.y, S.rx, S.ry};
  double *o = out + (size_t)q * ASL_SSM_NFEAT;
  const long long row = lib_rows[q];
  if (row < 0 || row >= L.n) {
    for (int f = lane; f < ASL_SSM_NFEAT; f += 64) o[f] = NAN;
    return;
  }
  const int qo = Qs.offsets[q], lo = L.offsets[row];
  int nq = Qs.offsets[q + 1] - qo, nl = L.offsets[row + 1] - lo;
  if (nq > MAXP || nl > MAXP) {
    if (lane == 0) atomicOr(status, 1);
    for (int f = lane; f < ASL_SSM_NFEAT; f += 64) o[f] = NAN;
    return;
  }
  int npm = pm_count[q];
  if (npm > pm_stride) npm = pm_stride;
  const uint32_t *pm = pm_pairs + (size_t)q * pm_stride * 2;
  for (int i = lane; i < MAXP; i += 64) {
    S.used_q[i] = 0;
    S.used_l[i] = 0;
    if (i < nq) {
      S.q_mz[i] = Qs.mz[qo + i];
      S.q_int[i] = Qs.intensity[qo + i];
    }
    if (i < nl) {
      S.l_mz[i] = L.mz[lo + i];
      S.l_int[i] = L.intensity[lo + i];
    }
  }
  sim_sync();
  // the `top` most intense library peaks (spectrum_similarity.py:51-53; ties: later peak first)
  for (int i = lane; i < nl; i += 64) {
    int above = 0;
    const float li = S.l_int[i];
    for (int j = 0; j < nl; j++) above += S.l_int[j] > li || (S.l_int[j] == li && j > i);
    S.in_top[i] = above < top;
  }
  bool bad = false;
  for (int i = lane; i < npm; i += 64) {
    const uint32_t a = pm[2 * i], b = pm[2 * i + 1];
    if (a >= (uint32_t)nq || b >= (uint32_t)nl) {
      bad = true;
      continue;
    }
    S.used_q[a] = 1;
    S.used_l[b] = 1;
    S.mq[i] = S.q_int[a];
    S.ml[i] = S.l_int[b];
    S.mzq[i] = S.q_mz[a];
    S.mzl[i] = S.l_mz[b];
  }
  if (__ballot(bad)) {
    if (lane == 0) atomicOr(status, 2);
    for (int f = lane; f < ASL_SSM_NFEAT; f += 64) o[f] = NAN;
    return;
  }
  sim_sync();
  const int n = npm;
  const int nt = compact(lane, n, [&](int i, int pos) {
    const bool f = S.in_top[pm[2 * i + 1]];
    if (pos >= 0) {
      S.tq[pos] = S.mq[i];
      S.tl[pos] = S.ml[i];
      S.tzq[pos] = S.mzq[i];
      S.tzl[pos] = S.mzl[i];
    }
    return f;
  });
  int n_uq = compact(lane, nq, [&](int i, int pos) {
    if (pos >= 0) S.uq[pos] = S.q_int[i];
    return !S.used_q[i];
  });
  int n_ul = compact(lane, nl, [&](int i, int pos) {
    if (pos >= 0) S.ul[pos] = S.l_int[i];
    return !S.used_l[i];
  });
  int n_tul = compact(lane, nl, [&](int i, int pos) {
    if (pos >= 0) S.tul[pos] = S.l_int[i];
    return !S.used_l[i] && S.in_top[i];
  });
  if (n == 0) n_ul = n_tul = 0;
  sim_sync();
  const SimView full{n, n_ul, S.mq, S.ml, S.mzq, S.mzl, S.ul};
  const SimView tv{nt, n_tul, S.tq, S.tl, S.tzq, S.tzl, S.tul};

  double sum_q = 0, sum_l = 0, s_mq = 0, s_ml = 0, s_uq = 0, s_ul = 0, s_tl = 0, s_tul = 0;
  double ssuq = 0, ssul = 0, muq = 0, mul = 0;
  long long nzuq = 0, nzul = 0;
  for (int i = lane; i < nq; i += 64) sum_q += S.q_int[i];
  for (int i = lane; i < nl; i += 64) sum_l += S.l_int[i];
  for (int i = lane; i < n; i += 64) {
    s_mq += S.mq[i];
    s_ml += S.ml[i];
  }
  for (int i = lane; i < n_uq; i += 64) {
    const double u = S.uq[i];
    s_uq += u;
    ssuq += u * u;
    muq = fmax(muq, u);
    nzuq += u != 0.0;
  }
  for (int i = lane; i < n_ul; i += 64) {
    const double u = S.ul[i];
    s_ul += u;
    ssul += u * u;
    mul = fmax(mul, u);
    nzul += u != 0.0;
  }
  for (int i = lane; i < nt; i += 64) s_tl += S.tl[i];
  for (int i = lane; i < n_tul; i += 64) s_tul += S.tul[i];
  sum_q = wsum(sum_q);
  sum_l = wsum(sum_l);
  s_mq = wsum(s_mq);
  s_ml = wsum(s_ml);
  s_uq = wsum(s_uq);
  s_ul = wsum(s_ul);
  s_tl = wsum(s_tl);
  s_tul = wsum(s_tul);
  ssuq = wsum(ssuq);
  ssul = wsum(ssul);
  muq = wmax(muq);
  mul = wmax(mul);
  nzuq = wsum_ll(nzuq);
  nzul = wsum_ll(nzul);

  double f[ASL_SSM_NFEAT];
  f[0] = cosine_of(lane, full, false);
  f[1] = cosine_of(lane, tv, true);
  f[2] = n;
  f[3] = n ? (double)n / nq : 0.0;
  f[4] = n ? (double)n / nl : 0.0;
  f[5] = nt ? (double)nt / (nt + n_tul) : 0.0;
  f[6] = n ? s_mq / sum_q : 0.0;
  f[7] = n ? s_ml / sum_l : 0.0;
  f[8] = nt ? s_tl / (s_tl + s_tul) : 0.0;
  f[9] = mse_of(lane, S.mzq, S.mzl, n);
  f[10] = mse_of(lane, S.tzq, S.tzl, nt);
  f[11] = mse_of(lane, S.mq, S.ml, n);
  f[12] = mse_of(lane, S.tq, S.tl, nt);
  f[13] = 1.0 - 2.0 * acos(fmax(0.0, fmin(1.0, f[0]))) / M_PI;
  f[14] = 1.0 - 2.0 * acos(fmax(0.0, fmin(1.0, f[1]))) / M_PI;
  {  // hypergeometric score, spectrum_similarity.py:251-306
    const double ldenom = lcomb(n_bins, nl);
    double prob = 0.0;
    for (int i = n + 1 + lane; i <= nl; i += 64) {
      const double l = lcomb(nl, i) + lcomb(n_bins - nl, nl - i) - ldenom;
      if (l > -INFINITY) prob += exp(l);
    }
    const double v = -log(wsum(prob));
    f[15] = v < 100.0 ? v : 100.0;
  }
  f[16] = n ? kendall_neglogp(lane, S.mq, S.ml, n, S.kc[0], S.kc[1]) : 0.0;
  double sad = 0, ssd = 0, maxd = 0, sadmz = 0, ssum = 0, smin = 0, smax = 0, canb = 0;
  for (int i = lane; i < n; i += 64) {
    const double d = fabs((double)(S.mq[i] - S.ml[i]));
    const double a = S.mq[i], b = S.ml[i];
    sad += d;
    ssd += d * d;
    maxd = fmax(maxd, d);
    sadmz += fabs((double)(S.mzq[i] - S.mzl[i]));
    ssum += fabs(a + b);
    smin += fmin(a, b);
    smax += fmax(a, b);
    const double c = d / (a + b);
    if (!isnan(c)) canb += isinf(c) ? 1.79769313486231570e308 : c;
  }
  sad = wsum(sad);
  ssd = wsum(ssd);
  maxd = wmax(maxd);
  sadmz = wsum(sadmz);
  ssum = wsum(ssum);
  smin = wsum(smin);
  smax = wsum(smax);
  canb = wsum(canb);
  const double n4 = (double)n * n * n * n;
  f[17] = n ? fmin(n4 / ((double)nq * nl * pow(fmax(sad, 2.220446049250313e-16), 0.25)), 1000.0) : 0.0;
  f[18] = n ? n4 * pow(sum_q + 2.0 * sum_l, 1.25) /
                  (((double)nq + 2.0 * nl) * ((double)nq + 2.0 * nl) + sad + sadmz)
            : 0.0;
  for (int w = 0; w < 2; w++) {   // spectral entropy, spectrum_similarity.py:653-700
    if (!n) {
      f[19 + w] = 0.0;
      continue;
    }
    sim_sync();
    for (int i = lane; i < nq; i += 64) S.x[i] = S.q_int[i];
    sim_sync();
    const double hq = spectrum_entropy(lane, S.x, nq, w);
    sim_sync();
    for (int i = lane; i < nl; i += 64) S.x[i] = S.l_int[i];
    sim_sync();
    const double hl = spectrum_entropy(lane, S.x, nl, w);
    sim_sync();
    for (int i = lane; i < n; i += 64) S.x[i] = ((double)S.mq[i] + (double)S.ml[i]) / 2.0;
    for (int i = lane; i < n_uq; i += 64) S.x[n + i] = (double)S.uq[i] / 2.0;
    for (int i = lane; i < n_ul; i += 64) S.x[n + n_uq + i] = (double)S.ul[i] / 2.0;
    sim_sync();
    const double hm = spectrum_entropy(lane, S.x, n + n_uq + n_ul, w);
    f[19 + w] = 1.0 - (2.0 * hm - hq - hl) / log(4.0);
  }
  f[21] = scribe_of(lane, full);
  f[22] = scribe_of(lane, tv);
  f[23] = n ? sad + s_uq + s_ul : INFINITY;
  f[24] = n ? sqrt(ssd + ssuq + ssul) : INFINITY;
  f[25] = n ? fmax(maxd, fmax(muq, mul)) : INFINITY;
  f[26] = corr_of(lane, full, false, scr);
  f[27] = corr_of(lane, tv, false, scr);
  f[28] = corr_of(lane, full, true, scr);
  f[29] = corr_of(lane, tv, true, scr);
  f[30] = n ? (sad + s_uq + s_ul) / (ssum + s_uq + s_ul) : 1.0;
  f[31] = n ? canb + (double)nzuq + (double)nzul : INFINITY;
  f[32] = n ? smin / (smax + s_uq + s_ul) : 0.0;
  if (lane == 0)
    for (int k = 0; k < ASL_SSM_NFEAT; k++) o[k] = f[k];
}

// Column 0 alone (the cosine the cascade uses as its default search-engine score,
// spectrum_similarity.py:81-106 / utils.py:407): one wave per SSM, the same lane-strided
// double accumulation and wave tree sum as cosine_of() above, so the value has the bits of
// features[:, 0]; no LDS staging, no peak-count limit. 33x less work per SSM.
__global__ __launch_bounds__(256) void ssm_cosine_kernel(
    DevPeaks Qs, DevPeaks L, const int32_t *__restrict__ lib_rows,
    const uint32_t *__restrict__ pm_pairs, const int32_t *__restrict__ pm_count, int pm_stride,
    double *__restrict__ out, int *status) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + wave;
  if (q >= Qs.n) return;
  const long long row = lib_rows[q];
  if (row < 0 || row >= L.n) {
    if (lane == 0) out[q] = NAN;
    return;
  }
  const int qo = Qs.offsets[q], lo = L.offsets[row];
  const int nq = Qs.offsets[q + 1] - qo, nl = L.offsets[row + 1] - lo;
  int n = pm_count[q];
  if (n > pm_stride) n = pm_stride;
  const uint32_t *pm = pm_pairs + (size_t)q * pm_stride * 2;
  double d = 0.0;
  bool bad = false;
  for (int i = lane; i < n; i += 64) {
    const uint32_t a = pm[2 * i], b = pm[2 * i + 1];
    if (a >= (uint32_t)nq || b >= (uint32_t)nl) {
      bad = true;
      continue;
    }
    d += (double)Qs.intensity[qo + a] * (double)L.intensity[lo + b];
  }
  if (__ballot(bad)) {
    if (lane == 0) {
      atomicOr(status, 2);
      out[q] = NAN;
    }
    return;
  }
  d = wsum(d);
  if (lane == 0) out[q] = n ? d : 0.0;
}

}  // namespace asl

using namespace asl;

extern "C" int asl_ssm_cosine_batch(const asl_peaks_t *queries, const asl_peaks_t *library,
                                    const int32_t *lib_rows, const uint32_t *pm_pairs,
                                    const int32_t *pm_count, int32_t pm_stride, double *cosine) {
  clear_error();
  if (!queries || !library || !lib_rows || !pm_count || !cosine)
    return fail(ASL_ERR_INVALID, "ssm_cosine: null argument");
  const int nq = queries->n;
  if (nq == 0) return ASL_OK;
  if (pm_stride <= 0 || !pm_pairs) return fail(ASL_ERR_INVALID, "ssm_cosine: pm_pairs/pm_stride");
  ASL_TRY(ensure_device());
  PeaksStage Q, L;
  ASL_TRY(Q.init(queries));
  ASL_TRY(L.init(library));
  In<int32_t> rows, cnt;
  In<uint32_t> pairs;
  Out<double> o;
  ASL_TRY(rows.init(lib_rows, nq));
  ASL_TRY(cnt.init(pm_count, nq));
  ASL_TRY(pairs.init(pm_pairs, (size_t)nq * pm_stride * 2));
  ASL_TRY(o.init(cosine, (size_t)nq));
  DevBuf<int> status;
  ASL_TRY(status.reserve(1));
  HIP_TRY(hipMemsetAsync(status.p, 0, sizeof(int), stream()));
  {
    ProfScope ps("ssm_cosine");
    hipLaunchKernelGGL(ssm_cosine_kernel, dim3((unsigned)cdiv(nq, 4)), dim3(256), 0, stream(),
                       Q.dev, L.dev, rows.d, pairs.d, cnt.d, pm_stride, o.d, status.p);
    ASL_CHECK_LAUNCH();
  }
  int st = 0;
  HIP_TRY(hipMemcpyAsync(&st, status.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
  ASL_TRY(o.finish());
  ASL_TRY(sync_stream());
  if (st & 2) return fail(ASL_ERR_INVALID, "ssm_cosine: a peak match index is out of range");
  return ASL_OK;
}

extern "C" int asl_ssm_features_batch(const asl_peaks_t *queries, const asl_peaks_t *library,
                                      const int32_t *lib_rows, const uint32_t *pm_pairs,
                                      const int32_t *pm_count, int32_t pm_stride, double min_mz,
                                      double max_mz, double bin_size, int32_t top,
                                      double *features) {
  clear_error();
  if (!queries || !library || !lib_rows || !pm_count || !features)
    return fail(ASL_ERR_INVALID, "ssm_features: null argument");
  const int nq = queries->n;
  if (nq == 0) return ASL_OK;
  if (pm_stride <= 0 || !pm_pairs) return fail(ASL_ERR_INVALID, "ssm_features: pm_pairs/pm_stride");
  if (top <= 0) return fail(ASL_ERR_INVALID, "ssm_features: top must be positive");
  ASL_TRY(ensure_device());
  int64_t n_bins = 0;
  double d0, d1;
  ASL_TRY(asl_get_dim(min_mz, max_mz, bin_size, &n_bins, &d0, &d1));
  PeaksStage Q, L;
  ASL_TRY(Q.init(queries));
  ASL_TRY(L.init(library));
  In<int32_t> rows, cnt;
  In<uint32_t> pairs;
  Out<double> o;
  ASL_TRY(rows.init(lib_rows, nq));
  ASL_TRY(cnt.init(pm_count, nq));
  ASL_TRY(pairs.init(pm_pairs, (size_t)nq * pm_stride * 2));
  ASL_TRY(o.init(features, (size_t)nq * ASL_SSM_NFEAT));
  DevBuf<int> status;
  ASL_TRY(status.reserve(1));
  HIP_TRY(hipMemsetAsync(status.p, 0, sizeof(int), stream()));
  int st = 0;
  {
    ProfScope ps("ssm_features");
    // 128-peak instantiation first (10 waves per CU); a batch with a longer spectrum is
    // redone with the 256-peak one
    hipLaunchKernelGGL(ssm_features_kernel<128>, dim3((unsigned)cdiv(nq, SIM_WAVES)),
                       dim3(64 * SIM_WAVES), 0, stream(), Q.dev, L.dev, rows.d, pairs.d, cnt.d,
                       pm_stride, (double)n_bins, top, o.d, status.p);
    ASL_CHECK_LAUNCH();
    HIP_TRY(hipMemcpyAsync(&st, status.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
    ASL_TRY(sync_stream());
    if (st & 1) {
      HIP_TRY(hipMemsetAsync(status.p, 0, sizeof(int), stream()));
      hipLaunchKernelGGL(ssm_features_kernel<SIM_MAXP>, dim3((unsigned)cdiv(nq, SIM_WAVES)),
                         dim3(64 * SIM_WAVES), 0, stream(), Q.dev, L.dev, rows.d, pairs.d, cnt.d,
                         pm_stride, (double)n_bins, top, o.d, status.p);
      ASL_CHECK_LAUNCH();
      HIP_TRY(hipMemcpyAsync(&st, status.p, sizeof(int), hipMemcpyDeviceToHost, stream()));
      ASL_TRY(sync_stream());
    }
  }
  ASL_TRY(o.finish());
  ASL_TRY(sync_stream());
  if (st & 1) return fail(ASL_ERR_CAPACITY, "ssm_features: a spectrum has more than %d peaks", SIM_MAXP);
  if (st & 2) return fail(ASL_ERR_INVALID, "ssm_features: a peak match index is out of range");
  return ASL_OK;
}

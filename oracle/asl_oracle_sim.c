/* asl_oracle_sim.c -- TEST INFRASTRUCTURE (see asl_oracle.h): CPU restatement of the
 * similarity features the reference computes per spectrum-spectrum match
 * (/root/reference/src/ann_solo/spectrum_similarity.py:13-730, called from
 * /root/reference/src/ann_solo/utils.py:344-456). Pinned to the reference's own test
 * constants (src/tests/spectrum_similarity_test.py, tests/golden/similarity_expected.json)
 * and to tests/golden/ssm_features_golden.npz (the reference module run on seeded SSMs with
 * the scipy of the build container). scipy pieces restated from their published
 * definitions: stats.kendalltau (tau-b p-value: exact for no ties and n <= 33 or <= 1
 * discordant/concordant pair, else the tie-corrected normal approximation),
 * stats.pearsonr, stats.spearmanr (average ranks), stats.entropy, special.comb.
 *
 * Inputs are the float32 peak arrays; sums are carried in double (the reference sums in
 * float32 with NumPy's pairwise order: agreement is to ~1e-6 relative, the tests allow 1e-5). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "asl_oracle.h"

#define SIM_MAXP 4096

static double lcomb(double n, double k) { /* log C(n,k), -inf when k > n or k < 0 */
  if (k < 0 || k > n) return -INFINITY;
  return lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0);
}

/* spectrum_similarity.py:251-306 */
static double hypergeometric(int n_matched, int n_library, double n_bins) {
  double prob = 0.0;
  const double ldenom = lcomb(n_bins, n_library);
  for (int i = n_matched + 1; i <= n_library; i++) {
    const double l = lcomb(n_library, i) + lcomb(n_bins - n_library, n_library - i) - ldenom;
    if (l > -INFINITY) prob += exp(l);
  }
  const double v = -log(prob);
  return v < 100.0 ? v : 100.0;
}

/* scipy.stats.entropy of a non-negative vector (natural log), entr(0) = 0 */
static double entropy_of(const double *x, int n) {
  double s = 0.0, h = 0.0;
  for (int i = 0; i < n; i++) s += x[i];
  for (int i = 0; i < n; i++) {
    const double p = x[i] / s;
    if (p > 0.0) h -= p * log(p);
  }
  return h;
}

/* spectrum_similarity.py:703-730 */
static double spectrum_entropy(const double *x, int n, int weighted, double *tmp) {
  const double h = entropy_of(x, n);
  if (!weighted || h > 3.0) return h;
  const double w = 0.25 + (1.0 - 0.25) / 3.0 * h;
  for (int i = 0; i < n; i++) tmp[i] = pow(x[i], w);
  return entropy_of(tmp, n);
}

/* scipy.stats.rankdata(method='average') */
static void avg_ranks(const double *x, int n, double *r) {
  for (int i = 0; i < n; i++) {
    int less = 0, eq = 0;
    for (int j = 0; j < n; j++) {
      less += x[j] < x[i];
      eq += x[j] == x[i];
    }
    r[i] = less + 0.5 * (eq + 1);
  }
}

/* scipy.stats.pearsonr statistic; NaN (constant input) -> 0 as the reference maps it */
static double pearson(const double *x, const double *y, int n) {
  if (n < 2) return 0.0;
  int cx = 1, cy = 1;
  double mx = 0, my = 0;
  for (int i = 0; i < n; i++) {
    cx &= x[i] == x[0];
    cy &= y[i] == y[0];
    mx += x[i];
    my += y[i];
  }
  if (cx || cy) return 0.0;
  mx /= n;
  my /= n;
  double xmax = 0, ymax = 0;
  for (int i = 0; i < n; i++) {
    xmax = fmax(xmax, fabs(x[i] - mx));
    ymax = fmax(ymax, fabs(y[i] - my));
  }
  double sx = 0, sy = 0;
  for (int i = 0; i < n; i++) {
    sx += ((x[i] - mx) / xmax) * ((x[i] - mx) / xmax);
    sy += ((y[i] - my) / ymax) * ((y[i] - my) / ymax);
  }
  const double nx = xmax * sqrt(sx), ny = ymax * sqrt(sy);
  double r = 0;
  for (int i = 0; i < n; i++) r += (x[i] - mx) / nx * (y[i] - my) / ny;
  r = fmax(-1.0, fmin(1.0, r));
  if (n == 2) r = round(r);
  return isnan(r) ? 0.0 : r;
}

/* -log(p) of scipy.stats.kendalltau(x, y) (variant b, method 'auto', two-sided) */
static double kendall_neglogp(const double *x, const double *y, int n) {
  if (n < 2) return 0.0;
  long long dis = 0, xtie = 0, ytie = 0, ntie = 0;
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      const int ex = x[i] == x[j], ey = y[i] == y[j];
      xtie += ex;
      ytie += ey;
      ntie += ex && ey;
      if (!ex && !ey && ((x[i] < x[j]) != (y[i] < y[j]))) dis++;
    }
  const long long tot = (long long)n * (n - 1) / 2;
  if (xtie == tot || ytie == tot) return 0.0; /* NaN p-value -> 0 */
  const long long con_minus_dis = tot - xtie - ytie + ntie - 2 * dis;
  double p;
  const long long mn = dis < tot - dis ? dis : tot - dis;
  if (xtie == 0 && ytie == 0 && (n <= 33 || mn <= 1)) {
    /* _kendall_p_exact(n, c = tot - dis) */
    long long c = tot - dis;
    if (tot - c < c) c = tot - c;
    if (n == 1)
      p = 1.0;
    else if (n == 2)
      p = 1.0;
    else if (c == 0)
      p = 2.0 / tgamma(n + 1.0);
    else if (c == 1)
      p = 2.0 / tgamma((double)n);
    else if (4 * c == (long long)n * (n - 1))
      p = 1.0;
    else {
      double *cur = (double *)calloc((size_t)c + 1, sizeof(double));
      double *nxt = (double *)calloc((size_t)c + 1, sizeof(double));
      cur[0] = cur[1] = 1.0;
      for (int j = 3; j <= n; j++) {
        double acc = 0.0;
        for (long long i = 0; i <= c; i++) {
          acc += cur[i];
          nxt[i] = acc;
        }
        if (j <= c)
          for (long long i = c; i >= j; i--) nxt[i] -= nxt[i - j];
        double *t = cur;
        cur = nxt;
        nxt = t;
      }
      double s = 0.0;
      for (long long i = 0; i <= c; i++) s += cur[i];
      p = 2.0 * s / tgamma(n + 1.0);
      free(cur);
      free(nxt);
    }
    p = fmax(0.0, fmin(1.0, p));
  } else {
    /* tie statistics: sums over tie groups of t(t-1)(t-2) and t(t-1)(2t+5) */
    double x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    for (int i = 0; i < n; i++) {
      int firstx = 1, firsty = 1, tx = 0, ty = 0;
      for (int j = 0; j < n; j++) {
        if (x[j] == x[i]) {
          if (j < i) firstx = 0;
          tx++;
        }
        if (y[j] == y[i]) {
          if (j < i) firsty = 0;
          ty++;
        }
      }
      if (firstx && tx > 1) {
        x0 += (double)tx * (tx - 1.0) * (tx - 2.0);
        x1 += (double)tx * (tx - 1.0) * (2.0 * tx + 5.0);
      }
      if (firsty && ty > 1) {
        y0 += (double)ty * (ty - 1.0) * (ty - 2.0);
        y1 += (double)ty * (ty - 1.0) * (2.0 * ty + 5.0);
      }
    }
    const double m = (double)n * (n - 1.0);
    const double var = (m * (2.0 * n + 5.0) - x1 - y1) / 18.0 + (2.0 * xtie * ytie) / m +
                       x0 * y0 / (9.0 * m * (n - 2.0));
    const double z = (double)con_minus_dis / sqrt(var);
    p = erfc(fabs(z) / sqrt(2.0)); /* 2 * sf(|z|) */
  }
  if (isnan(p)) return 0.0;
  const double v = -log(p);
  return v == 0.0 ? 0.0 : v;
}

typedef struct {
  int n;                 /* matched pairs (after the top filter) or 0 == "None" */
  double *mq, *ml;       /* matched intensities */
  double *mzq, *mzl;     /* matched m/z */
  double *ul;            /* unmatched library intensities (restricted to top) */
  int n_ul;
} sim_view_t;

static double cosine_of(const sim_view_t *v, int recalculate_norm) {
  if (!v->n) return 0.0;
  double d = 0, a = 0, b = 0;
  for (int i = 0; i < v->n; i++) {
    d += v->mq[i] * v->ml[i];
    a += v->mq[i] * v->mq[i];
    b += v->ml[i] * v->ml[i];
  }
  return recalculate_norm ? d / (sqrt(a) * sqrt(b)) : d;
}

static double mse_of(const double *a, const double *b, int n) {
  if (!n) return INFINITY;
  double s = 0;
  for (int i = 0; i < n; i++) {
    const double d = (double)(float)(a[i] - b[i]); /* float32 subtraction in the reference */
    s += d * d;
  }
  return s / n;
}

static double scribe_of(const sim_view_t *v) {
  if (!v->n) return 0.0;
  double den = 0;
  for (int i = 0; i < v->n; i++) {
    const double d = (double)(float)(v->mq[i] - v->ml[i]);
    den += d * d;
  }
  for (int i = 0; i < v->n_ul; i++) den += v->ul[i] * v->ul[i];
  return den == 0.0 ? 10.0 : log(1.0 / den);
}

static double corr_of(const sim_view_t *v, int spearman, double *x, double *y, double *rx,
                      double *ry) {
  if (!v->n) return 0.0;
  const int n = v->n + v->n_ul;
  for (int i = 0; i < v->n; i++) {
    x[i] = v->mq[i];
    y[i] = v->ml[i];
  }
  for (int i = 0; i < v->n_ul; i++) {
    x[v->n + i] = 0.0;
    y[v->n + i] = v->ul[i];
  }
  if (!spearman) return pearson(x, y, n);
  avg_ranks(x, n, rx);
  avg_ranks(y, n, ry);
  return pearson(rx, ry, n);
}

/* out[ORC_SIM_NFEAT]: the similarity columns of utils.py:296-343 in dictionary order:
 *  0 cosine 1 cosine_top 2 n_matched_peaks 3 frac_n_peaks_query 4 frac_n_peaks_lib
 *  5 frac_n_peaks_lib_top 6 frac_int_query 7 frac_int_lib 8 frac_int_lib_top 9 mse_mz
 * 10 mse_mz_top 11 mse_int 12 mse_int_top 13 contrast_angle 14 contrast_angle_top
 * 15 hypergeometric_score 16 kendalltau 17 ms_for_id_v1 18 ms_for_id_v2
 * 19 entropy_unweighted 20 entropy_weighted 21 scribe_fragment_acc 22 scribe_fragment_acc_top
 * 23 manhattan 24 euclidean 25 chebyshev 26 pearsonr 27 pearsonr_top 28 spearmanr
 * 29 spearmanr_top 30 braycurtis 31 canberra 32 ruzicka */
void orc_ssm_features(const float *q_mz, const float *q_int, int32_t nq, const float *l_mz,
                      const float *l_int, int32_t nl, const uint32_t *pm, int32_t npm,
                      double min_mz, double max_mz, double bin_size, int32_t top, double *out) {
  const int cap = (nq > nl ? nq : nl) + 8;
  double *buf = (double *)calloc((size_t)cap * 16, sizeof(double));
  double *mq = buf, *ml = mq + cap, *mzq = ml + cap, *mzl = mzq + cap, *ul = mzl + cap;
  double *uq = ul + cap, *tq = uq + cap, *tl = tq + cap, *tzq = tl + cap, *tzl = tzq + cap;
  double *tul = tzl + cap, *x = tul + cap, *y = x + cap, *rx = y + cap, *ry = rx + cap;
  double *tmp = ry + cap;
  uint8_t *used_q = (uint8_t *)calloc((size_t)nq + 1, 1), *used_l = (uint8_t *)calloc((size_t)nl + 1, 1);
  uint8_t *in_top = (uint8_t *)calloc((size_t)nl + 1, 1);

  /* the `top` most intense library peaks (spectrum_similarity.py:51-53; argpartition leaves
   * the choice among equal intensities open -- here the later peak wins) */
  for (int i = 0; i < nl; i++) {
    int above = 0;
    for (int j = 0; j < nl; j++) above += l_int[j] > l_int[i] || (l_int[j] == l_int[i] && j > i);
    in_top[i] = above < top;
  }
  sim_view_t full = {0, mq, ml, mzq, mzl, ul, 0}, tv = {0, tq, tl, tzq, tzl, tul, 0};
  for (int i = 0; i < npm; i++) {
    const uint32_t a = pm[2 * i], b = pm[2 * i + 1];
    used_q[a] = used_l[b] = 1;
    mq[full.n] = q_int[a];
    ml[full.n] = l_int[b];
    mzq[full.n] = q_mz[a];
    mzl[full.n] = l_mz[b];
    full.n++;
    if (in_top[b]) {
      tq[tv.n] = q_int[a];
      tl[tv.n] = l_int[b];
      tzq[tv.n] = q_mz[a];
      tzl[tv.n] = l_mz[b];
      tv.n++;
    }
  }
  int n_uq = 0;
  for (int i = 0; i < nq; i++)
    if (!used_q[i]) uq[n_uq++] = q_int[i];
  for (int i = 0; i < nl; i++)
    if (!used_l[i]) {
      ul[full.n_ul++] = l_int[i];
      if (in_top[i]) tul[tv.n_ul++] = l_int[i];
    }
  if (npm == 0) full.n_ul = tv.n_ul = 0;

  const int n = full.n;
  double sum_q = 0, sum_l = 0, s_mq = 0, s_ml = 0, s_uq = 0, s_ul = 0, s_tl = 0, s_tul = 0;
  for (int i = 0; i < nq; i++) sum_q += q_int[i];
  for (int i = 0; i < nl; i++) sum_l += l_int[i];
  for (int i = 0; i < n; i++) {
    s_mq += mq[i];
    s_ml += ml[i];
  }
  for (int i = 0; i < n_uq; i++) s_uq += uq[i];
  for (int i = 0; i < full.n_ul; i++) s_ul += ul[i];
  for (int i = 0; i < tv.n; i++) s_tl += tl[i];
  for (int i = 0; i < tv.n_ul; i++) s_tul += tul[i];

  out[0] = cosine_of(&full, 0);
  out[1] = cosine_of(&tv, 1);
  out[2] = n;
  out[3] = n ? (double)n / nq : 0.0;
  out[4] = n ? (double)n / nl : 0.0;
  out[5] = tv.n ? (double)tv.n / (tv.n + tv.n_ul) : 0.0;
  out[6] = n ? s_mq / sum_q : 0.0;
  out[7] = n ? s_ml / sum_l : 0.0;
  out[8] = tv.n ? s_tl / (s_tl + s_tul) : 0.0;
  out[9] = mse_of(mzq, mzl, n);
  out[10] = mse_of(tzq, tzl, tv.n);
  out[11] = mse_of(mq, ml, n);
  out[12] = mse_of(tq, tl, tv.n);
  out[13] = 1.0 - 2.0 * acos(fmax(0.0, fmin(1.0, out[0]))) / M_PI;
  out[14] = 1.0 - 2.0 * acos(fmax(0.0, fmin(1.0, out[1]))) / M_PI;
  int64_t n_bins;
  double d0, d1;
  orc_get_dim(min_mz, max_mz, bin_size, &n_bins, &d0, &d1);
  out[15] = hypergeometric(n, nl, (double)n_bins);
  out[16] = n ? kendall_neglogp(mq, ml, n) : 0.0;

  double sad = 0, ssd = 0, maxd = 0, sadmz = 0, ssum = 0, smin = 0, smax = 0, canb = 0;
  for (int i = 0; i < n; i++) {
    const double d = fabs((double)(float)(mq[i] - ml[i]));
    sad += d;
    ssd += d * d;
    maxd = fmax(maxd, d);
    sadmz += fabs((double)(float)(mzq[i] - mzl[i]));
    ssum += fabs(mq[i] + ml[i]);
    smin += fmin(mq[i], ml[i]);
    smax += fmax(mq[i], ml[i]);
    const double c = d / (mq[i] + ml[i]);
    if (!isnan(c)) canb += isinf(c) ? 1.79769313486231570e308 : c;
  }
  out[17] = n ? fmin(pow((double)n, 4) / ((double)nq * nl * pow(fmax(sad, 2.220446049250313e-16), 0.25)),
                     1000.0)
              : 0.0;
  out[18] = n ? pow((double)n, 4) * pow(sum_q + 2.0 * sum_l, 1.25) /
                    (pow((double)nq + 2.0 * nl, 2) + sad + sadmz)
              : 0.0;
  for (int w = 0; w < 2; w++) {
    if (!n) {
      out[19 + w] = 0.0;
      continue;
    }
    for (int i = 0; i < nq; i++) x[i] = q_int[i];
    const double hq = spectrum_entropy(x, nq, w, tmp);
    for (int i = 0; i < nl; i++) x[i] = l_int[i];
    const double hl = spectrum_entropy(x, nl, w, tmp);
    /* merged spectrum needs nq + nl - n <= 2*cap entries: reuse x..ry (4*cap contiguous) */
    int k = 0;
    for (int i = 0; i < n; i++) x[k++] = (mq[i] + ml[i]) / 2.0;
    for (int i = 0; i < n_uq; i++) x[k++] = uq[i] / 2.0;
    for (int i = 0; i < full.n_ul; i++) x[k++] = ul[i] / 2.0;
    const double hm = spectrum_entropy(x, k, w, x + 2 * cap);
    out[19 + w] = 1.0 - (2.0 * hm - hq - hl) / log(4.0);
  }
  out[21] = scribe_of(&full);
  out[22] = scribe_of(&tv);
  double ssuq = 0, ssul = 0, muq = 0, mul = 0;
  int nzuq = 0, nzul = 0;
  for (int i = 0; i < n_uq; i++) {
    ssuq += uq[i] * uq[i];
    muq = fmax(muq, uq[i]);
    nzuq += uq[i] != 0.0;
  }
  for (int i = 0; i < full.n_ul; i++) {
    ssul += ul[i] * ul[i];
    mul = fmax(mul, ul[i]);
    nzul += ul[i] != 0.0;
  }
  out[23] = n ? sad + s_uq + s_ul : INFINITY;
  out[24] = n ? sqrt(ssd + ssuq + ssul) : INFINITY;
  out[25] = n ? fmax(maxd, fmax(muq, mul)) : INFINITY;
  out[26] = corr_of(&full, 0, x, y, rx, ry);
  out[27] = corr_of(&tv, 0, x, y, rx, ry);
  out[28] = corr_of(&full, 1, x, y, rx, ry);
  out[29] = corr_of(&tv, 1, x, y, rx, ry);
  out[30] = n ? (sad + s_uq + s_ul) / (ssum + s_uq + s_ul) : 1.0;
  out[31] = n ? canb + nzuq + nzul : INFINITY;
  out[32] = n ? smin / (smax + s_uq + s_ul) : 0.0;
  free(buf);
  free(used_q);
  free(used_l);
  free(in_top);
}

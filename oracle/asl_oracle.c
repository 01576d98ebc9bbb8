/*
 * asl_oracle.c -- CPU ORACLE (test infrastructure, see asl_oracle.h).
 *
 * Plain-C restatement of the ANN-SoLo hot path. Every function cites the
 * reference file:line it follows (paths relative to /root/reference).
 * Floating-point conventions that the HIP kernels share (DESIGN.md "Canonical
 * arithmetic"):
 *   - inner product  = ascending-k fp32 fmaf chain from +0
 *   - L2 distance    = ascending-k chain acc = fmaf(x-c, x-c, acc)
 *   - ADC sum        = p_j = sum_t LUT[j+16t] (t ascending), then the fixed
 *                      16->1 mirror tree (j,15-j)(j,7-j)(j,3-j)(0,1), then + coarse
 *   - top-k order    = (score desc, id asc)
 */
#define _GNU_SOURCE
#include "asl_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(__x86_64__)
#define ORC_CLONES __attribute__((target_clones("arch=haswell", "default")))
#else
#define ORC_CLONES
#endif

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* Encoder                                                                   */
/* ------------------------------------------------------------------------ */

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* MurmurHash3_x86_32 (public domain, Austin Appleby) -- what mmh3.hash()
 * computes; call site src/ann_solo/spectrum.py:163. */
uint32_t orc_murmur3_32(const uint8_t *key, int len, uint32_t seed) {
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  uint32_t h1 = seed;
  const int nblocks = len / 4;
  for (int i = 0; i < nblocks; i++) {
    uint32_t k1 = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) |
                  ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
  }
  const uint8_t *tail = key + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k1 ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1:
      k1 ^= tail[0];
      k1 *= c1;
      k1 = rotl32(k1, 15);
      k1 *= c2;
      h1 ^= k1;
  }
  h1 ^= (uint32_t)len;
  h1 ^= h1 >> 16;
  h1 *= 0x85ebca6bu;
  h1 ^= h1 >> 13;
  h1 *= 0xc2b2ae35u;
  h1 ^= h1 >> 16;
  return h1;
}

/* spectrum.py:146-163: mmh3.hash(str(bin_idx), 42, signed=False) % hash_len */
int32_t orc_hash_idx(int64_t bin_idx, int32_t hash_len, uint32_t seed) {
  char buf[32];
  int len = snprintf(buf, sizeof buf, "%lld", (long long)bin_idx);
  return (int32_t)(orc_murmur3_32((const uint8_t *)buf, len, seed) % (uint32_t)hash_len);
}

static double py_mod(double a, double b) { /* Python float % for b>0 */
  double m = fmod(a, b);
  if (m != 0.0 && ((b < 0) != (m < 0))) m += b;
  return m;
}

/* spectrum.py:122-143 */
void orc_get_dim(double min_mz, double max_mz, double bin_size, int64_t *n_bins,
                 double *start_dim, double *end_dim) {
  double s = min_mz - py_mod(min_mz, bin_size);
  double e = max_mz + bin_size - py_mod(max_mz, bin_size);
  if (start_dim) *start_dim = s;
  if (end_dim) *end_dim = e;
  if (n_bins) *n_bins = (int64_t)nearbyint((e - s) / bin_size); /* round-half-even */
}

/* NumPy's npy_floor_divide for doubles (what `//` does on the np.float64
 * scalar at spectrum.py:207). */
double orc_npy_floor_divide(double a, double b) {
  if (b == 0.0) return a / b;
  double mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.0) {
    if ((b < 0) != (mod < 0)) div -= 1.0;
  }
  double floordiv;
  if (div != 0.0) {
    floordiv = floor(div);
    if (div - floordiv > 0.5) floordiv += 1.0;
  } else {
    floordiv = copysign(0.0, a / b);
  }
  return floordiv;
}

/* spectrum.py:207 -- float64 arithmetic on the float32 m/z value. */
int64_t orc_bin_idx(float mz, double min_bound, double bin_size) {
  return (int64_t)floor(orc_npy_floor_divide((double)mz - min_bound, bin_size));
}

/* spectrum.py:166-214. fp32 scatter-add in peak order, then fp32 L2 norm
 * (ascending-index fmaf chain, sqrtf, IEEE divide). */
void orc_encode(const float *mz, const float *intensity, int32_t n_peaks, double min_bound,
                double bin_size, int32_t hash_len, uint32_t seed, int norm, float *out) {
  memset(out, 0, sizeof(float) * (size_t)hash_len);
  for (int32_t p = 0; p < n_peaks; p++) {
    int64_t b = orc_bin_idx(mz[p], min_bound, bin_size);
    int32_t h = orc_hash_idx(b, hash_len, seed);
    out[h] += intensity[p];
  }
  if (norm) {
    float acc = 0.0f;
    for (int32_t i = 0; i < hash_len; i++) acc = fmaf(out[i], out[i], acc);
    float nrm = sqrtf(acc);
    for (int32_t i = 0; i < hash_len; i++) out[i] = out[i] / nrm;
  }
}

void orc_encode_batch(const float *mz, const float *intensity, const int32_t *offsets,
                      int32_t n, double min_bound, double bin_size, int32_t hash_len,
                      uint32_t seed, int norm, float *out) {
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < n; i++)
    orc_encode(mz + offsets[i], intensity + offsets[i], offsets[i + 1] - offsets[i],
               min_bound, bin_size, hash_len, seed, norm, out + (size_t)i * hash_len);
}

/* ------------------------------------------------------------------------ */
/* process_spectrum: spectrum.py:57-119 over spectrum_utils 0.3.x             */
/* ------------------------------------------------------------------------ */

typedef struct {
  float inten;
  int32_t idx;
} orc_pk_t;

/* (intensity desc, index desc): the order a stable ascending argsort read from its tail gives */
static int pk_cmp(const void *a, const void *b) {
  const orc_pk_t *x = (const orc_pk_t *)a, *y = (const orc_pk_t *)b;
  if (x->inten != y->inten) return x->inten > y->inten ? -1 : 1;
  return x->idx > y->idx ? -1 : (x->idx < y->idx);
}

static int spectrum_ok(const float *mz, const uint8_t *keep, int32_t n, int min_peaks,
                       double min_range) { /* spectrum.py:13-36 */
  int cnt = 0, first = -1, last = -1;
  for (int32_t i = 0; i < n; i++)
    if (keep[i]) {
      if (first < 0) first = i;
      last = i;
      cnt++;
    }
  return cnt >= min_peaks && cnt > 0 && (double)(mz[last] - mz[first]) >= min_range;
}

/* spectrum_utils 0.3.x MsmsSpectrum.round(decimals, 'sum') (spectrum.py:84-85). The library
 * function is numba-compiled: np.round_ on a float is evaluated in double --
 * rint(x * 10^d) / 10^d for d >= 0, rint(x / 10^-d) * 10^-d for d < 0, ties to even -- and
 * stored back as float32. Peaks whose rounded m/z coincide are merged: intensity = sequential
 * float32 sum in m/z order, annotation (here: the source index) = that of the most intense
 * merged peak, first on ties (np.argmax). */
float orc_round_mz(float mz, int32_t decimals) {
  double x = (double)mz, p = 1.0;
  int32_t a = decimals < 0 ? -decimals : decimals;
  for (int32_t i = 0; i < a; i++) p *= 10.0;
  double y = decimals >= 0 ? rint(x * p) / p : rint(x / p) * p;
  return (float)y;
}

int orc_process_spectrum(const float *mz_in, const float *intensity_in, int32_t n,
                         double precursor_mz, int32_t precursor_charge,
                         const orc_process_params_t *p, float *out_mz, float *out_int,
                         int32_t *out_src, int32_t *n_out) {
  *n_out = 0;
  if (n <= 0) return 0;
  uint8_t *keep = (uint8_t *)malloc((size_t)n);
  float *mz = (float *)malloc(sizeof(float) * (size_t)n);
  float *intensity = (float *)malloc(sizeof(float) * (size_t)n);
  int32_t *src = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  orc_pk_t *pk = NULL;
  float *val = NULL;
  int ok = 0;
  memcpy(mz, mz_in, sizeof(float) * (size_t)n);
  memcpy(intensity, intensity_in, sizeof(float) * (size_t)n);
  for (int32_t i = 0; i < n; i++) src[i] = i;
  /* spectrum.py:79 set_mz_range: inclusive on both sides */
  for (int32_t i = 0; i < n; i++) keep[i] = (double)mz[i] >= p->min_mz && (double)mz[i] <= p->max_mz;
  if (!spectrum_ok(mz, keep, n, p->min_peaks, p->min_mz_range)) goto done;
  /* spectrum.py:84-85 round(resolution, 'sum') */
  if (p->round_mz) {
    for (int32_t i = 0; i < n; i++)
      if (keep[i]) mz[i] = orc_round_mz(mz[i], p->resolution);
    int32_t i = 0;
    while (i < n) {
      if (!keep[i]) {
        i++;
        continue;
      }
      int32_t j = i + 1, best = i;
      float sum = intensity[i];
      while (j < n && keep[j] && mz[j] == mz[i]) {
        sum += intensity[j];
        if (intensity[j] > intensity[best]) best = j;
        keep[j] = 0;
        j++;
      }
      intensity[i] = sum;
      src[i] = best;
      i = j;
    }
    if (!spectrum_ok(mz, keep, n, p->min_peaks, p->min_mz_range)) goto done;
  }
  /* spectrum.py:90-92 remove_precursor_peak(tol, 'Da', 2) */
  if (p->remove_precursor) {
    const double adduct = 1.0072766;
    double neutral = (precursor_mz - adduct) * (double)precursor_charge;
    for (int charge = precursor_charge; charge >= 1; charge--)
      for (int iso = 0; iso <= 2; iso++) {
        double rm = (neutral + iso) / charge + adduct;
        for (int32_t i = 0; i < n; i++)
          if (keep[i] && fabs((double)mz[i] - rm) <= p->remove_precursor_tolerance) keep[i] = 0;
      }
    if (!spectrum_ok(mz, keep, n, p->min_peaks, p->min_mz_range)) goto done;
  }
  /* spectrum.py:97-99 filter_intensity: strictly above min_intensity * max, top max_peaks */
  pk = (orc_pk_t *)malloc(sizeof(orc_pk_t) * (size_t)n);
  int m = 0;
  for (int32_t i = 0; i < n; i++)
    if (keep[i]) {
      pk[m].inten = intensity[i];
      pk[m].idx = i;
      m++;
    }
  qsort(pk, (size_t)m, sizeof(orc_pk_t), pk_cmp);
  double thresh = p->min_intensity * (double)pk[0].inten;
  memset(keep, 0, (size_t)n);
  int kept = 0;
  for (int r = 0; r < m && r < p->max_peaks; r++)
    if ((double)pk[r].inten > thresh) {
      keep[pk[r].idx] = 1;
      kept++;
    }
  if (!spectrum_ok(mz, keep, n, p->min_peaks, p->min_mz_range)) goto done;
  /* spectrum.py:104-110 scale_intensity; :112 L2 norm (canonical ascending fmaf chain) */
  val = (float *)calloc((size_t)n, sizeof(float));
  for (int r = 0; r < kept; r++) {
    int32_t i = pk[r].idx;
    if (p->scaling == 1)
      val[i] = (float)(p->max_peaks - r);
    else if (p->scaling == 2)
      val[i] = sqrtf(intensity[i]);
    else
      val[i] = intensity[i];
  }
  float acc = 0.0f;
  for (int32_t i = 0; i < n; i++)
    if (keep[i]) acc = fmaf(val[i], val[i], acc);
  float nrm = sqrtf(acc);
  int o = 0;
  for (int32_t i = 0; i < n; i++)
    if (keep[i]) {
      out_mz[o] = mz[i];
      out_int[o] = val[i] / nrm;
      out_src[o] = src[i];
      o++;
    }
  *n_out = o;
  ok = 1;
done:
  free(val);
  free(keep);
  free(pk);
  free(mz);
  free(intensity);
  free(src);
  return ok;
}

/* ------------------------------------------------------------------------ */
/* Rescoring: SpectrumMatch.cpp:8-133                                        */
/* ------------------------------------------------------------------------ */

typedef struct {
  float prod;
  uint32_t qi, ci;
} orc_match_t;

typedef struct {
  orc_match_t *m;
  int cap;
  uint8_t *qused, *cused;
  int ucap;
} orc_scratch_t;

static _Thread_local orc_scratch_t tls_scratch;

static void scratch_reserve(orc_scratch_t *s, int n) {
  if (n > s->cap) {
    int nc = s->cap ? s->cap : 256;
    while (nc < n) nc *= 2;
    s->m = (orc_match_t *)realloc(s->m, sizeof(orc_match_t) * (size_t)nc);
    s->cap = nc;
  }
}

double orc_dot_pair(const float *q_mz, const float *q_int, int32_t q_n, double q_pmz,
                    const float *c_mz, const float *c_int, const uint8_t *c_chg,
                    int32_t c_n, double c_pmz, int32_t c_charge, double tol,
                    int allow_shift, uint32_t *matches_out, int32_t *n_matches_out) {
  orc_scratch_t *S = &tls_scratch;
  if (n_matches_out) *n_matches_out = 0;
  if (c_n <= 0 || q_n <= 0) return 0.0; /* reference would underflow (cpp:41); unreachable there */
  /* cpp:18-31 */
  double pmd = (q_pmz - c_pmz) * (double)(uint32_t)c_charge;
  int num_shifts = (allow_shift && fabs(pmd) >= tol) ? c_charge + 1 : 1;
  enum { MAXS = 64 };
  if (num_shifts > MAXS) num_shifts = MAXS;
  uint32_t cursor[MAXS];
  double mass_diff[MAXS];
  for (int s = 0; s < num_shifts; s++) cursor[s] = 0;
  mass_diff[0] = 0.0;
  for (int s = 1; s < num_shifts; s++) mass_diff[s] = pmd / (double)s;

  int M = 0;
  /* cpp:35-87 */
  for (int32_t qi = 0; qi < q_n; qi++) {
    float qm = q_mz[qi];
    for (int s = 0; s < num_shifts; s++)
      while (cursor[s] < (uint32_t)c_n - 1 &&
             (double)qm - tol > (double)c_mz[cursor[s]] + mass_diff[s])
        cursor[s]++;
    for (int s = 0; s < num_shifts; s++) {
      for (uint32_t idx = 0;
           cursor[s] + idx < (uint32_t)c_n &&
           fabs((double)qm - ((double)c_mz[cursor[s] + idx] + mass_diff[s])) <= tol;
           idx++) {
        uint32_t ci = cursor[s] + idx;
        double mult = 0.0;
        if (s == 0)
          mult = 1.0;
        else if (c_chg[ci] == s)
          mult = 1.0;
        else if (c_chg[ci] == 0)
          mult = 2.0 / 3.0;
        if (mult > 0.0) {
          scratch_reserve(S, M + 1);
          /* cpp:81: double product rounded to float by the tuple<float,..> store */
          S->m[M].prod = (float)(mult * (double)q_int[qi] * (double)c_int[ci]);
          S->m[M].qi = (uint32_t)qi;
          S->m[M].ci = ci;
          M++;
        }
      }
    }
  }
  /* cpp:92-93: sort by product, descending. std::sort is unstable; the oracle
   * (and the HIP kernel) define ties as generation order (stable). */
  for (int i = 1; i < M; i++) {
    orc_match_t key = S->m[i];
    int j = i - 1;
    while (j >= 0 && S->m[j].prod < key.prod) {
      S->m[j + 1] = S->m[j];
      j--;
    }
    S->m[j + 1] = key;
  }
  /* cpp:94-111 greedy one-to-one assignment */
  int need = q_n > c_n ? q_n : c_n;
  if (need > S->ucap) {
    S->qused = (uint8_t *)realloc(S->qused, (size_t)need);
    S->cused = (uint8_t *)realloc(S->cused, (size_t)need);
    S->ucap = need;
  }
  memset(S->qused, 0, (size_t)q_n);
  memset(S->cused, 0, (size_t)c_n);
  double score = 0.0;
  int nm = 0;
  for (int i = 0; i < M; i++) {
    uint32_t qi = S->m[i].qi, ci = S->m[i].ci;
    if (!S->qused[qi] && !S->cused[ci]) {
      score += (double)S->m[i].prod;
      if (matches_out) {
        matches_out[2 * nm] = qi;
        matches_out[2 * nm + 1] = ci;
      }
      nm++;
      S->qused[qi] = 1;
      S->cused[ci] = 1;
    }
  }
  if (n_matches_out) *n_matches_out = nm;
  return score;
}

int32_t orc_best_match(const orc_peaks_t *Q, int32_t qi, const orc_peaks_t *L,
                       const int64_t *cand_rows, int32_t n_cand, double tol,
                       int allow_shift, double *score_out, uint32_t *matches_out,
                       int32_t *n_matches_out) {
  int32_t qo = Q->offsets[qi], qn = Q->offsets[qi + 1] - qo;
  int32_t best = -1;
  double best_score = 0.0;
  /* cpp:13,118-129: first candidate always taken, later ones only if strictly better */
  for (int32_t c = 0; c < n_cand; c++) {
    int64_t r = cand_rows[c];
    int32_t co = L->offsets[r], cn = L->offsets[r + 1] - co;
    double s = orc_dot_pair(Q->mz + qo, Q->intensity + qo, qn, Q->precursor_mz[qi],
                            L->mz + co, L->intensity + co, L->charge + co, cn,
                            L->precursor_mz[r], L->precursor_charge[r], tol, allow_shift,
                            NULL, NULL);
    if (best < 0 || best_score < s) {
      best = c;
      best_score = s;
    }
  }
  if (best >= 0 && (matches_out || n_matches_out)) {
    int64_t r = cand_rows[best];
    int32_t co = L->offsets[r], cn = L->offsets[r + 1] - co;
    orc_dot_pair(Q->mz + qo, Q->intensity + qo, qn, Q->precursor_mz[qi], L->mz + co,
                 L->intensity + co, L->charge + co, cn, L->precursor_mz[r],
                 L->precursor_charge[r], tol, allow_shift, matches_out, n_matches_out);
  } else if (n_matches_out) {
    *n_matches_out = 0;
  }
  if (score_out) *score_out = best_score;
  return best;
}

/* ------------------------------------------------------------------------ */
/* Inner product, top-k                                                      */
/* ------------------------------------------------------------------------ */

ORC_CLONES float orc_ip(const float *a, const float *b, int32_t d) {
  float acc = 0.0f;
  for (int32_t k = 0; k < d; k++) acc = fmaf(a[k], b[k], acc);
  return acc;
}

/* same chain, skipping exact-zero query entries: fmaf(0,b,acc)==acc */
ORC_CLONES static float ip_sparse_q(const int32_t *idx, const float *val, int nnz,
                                    const float *b) {
  float acc = 0.0f;
  for (int t = 0; t < nnz; t++) acc = fmaf(val[t], b[idx[t]], acc);
  return acc;
}

ORC_CLONES static float l2_chain(const float *x, const float *c, int32_t d) {
  float acc = 0.0f;
  for (int32_t k = 0; k < d; k++) {
    float df = x[k] - c[k];
    acc = fmaf(df, df, acc);
  }
  return acc;
}

static int sparsify(const float *q, int32_t d, int32_t *idx, float *val) {
  int n = 0;
  for (int32_t k = 0; k < d; k++)
    if (q[k] != 0.0f) {
      idx[n] = k;
      val[n] = q[k];
      n++;
    }
  return n;
}

/* total order: better = higher score, then lower id */
typedef struct {
  float s;
  int64_t id;
} orc_hit_t;

static inline int hit_better(orc_hit_t a, orc_hit_t b) {
  return a.s > b.s || (a.s == b.s && a.id < b.id);
}

/* bounded "keep the k best" as a heap whose root is the WORST kept hit */
typedef struct {
  orc_hit_t *h;
  int n, k;
} orc_topk_t;

static void topk_sift_down(orc_hit_t *h, int n, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, w = i;
    if (l < n && hit_better(h[w], h[l])) w = l;
    if (r < n && hit_better(h[w], h[r])) w = r;
    if (w == i) break;
    orc_hit_t t = h[i];
    h[i] = h[w];
    h[w] = t;
    i = w;
  }
}

static inline void topk_push(orc_topk_t *t, float s, int64_t id) {
  orc_hit_t x = {s, id};
  if (t->n < t->k) {
    int i = t->n++;
    t->h[i] = x;
    while (i > 0) { /* sift up: parent must be worse-or-equal */
      int p = (i - 1) / 2;
      if (hit_better(t->h[p], t->h[i])) {
        orc_hit_t tmp = t->h[p];
        t->h[p] = t->h[i];
        t->h[i] = tmp;
        i = p;
      } else
        break;
    }
  } else if (t->k > 0 && hit_better(x, t->h[0])) {
    t->h[0] = x;
    topk_sift_down(t->h, t->n, 0);
  }
}

static int hit_cmp(const void *a, const void *b) {
  orc_hit_t x = *(const orc_hit_t *)a, y = *(const orc_hit_t *)b;
  if (hit_better(x, y)) return -1;
  if (hit_better(y, x)) return 1;
  return 0;
}

static void topk_finish(orc_topk_t *t, float *D, int64_t *I) {
  qsort(t->h, (size_t)t->n, sizeof(orc_hit_t), hit_cmp);
  for (int i = 0; i < t->k; i++) {
    if (i < t->n) {
      D[i] = t->h[i].s;
      I[i] = t->h[i].id;
    } else {
      D[i] = -FLT_MAX; /* FAISS pads D with the heap's neutral value, I with -1 */
      I[i] = -1;
    }
  }
}

/* IndexFlatIP.search (notebooks/iprg2012_num_candidates.ipynb:175): exact. */
void orc_flat_search(const float *xb, int64_t nb, const float *xq, int32_t nq, int32_t d,
                     int32_t k, float *D, int64_t *I) {
#pragma omp parallel
  {
    orc_hit_t *heap = (orc_hit_t *)malloc(sizeof(orc_hit_t) * (size_t)(k > 0 ? k : 1));
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)d);
    float *val = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(dynamic, 4)
    for (int32_t q = 0; q < nq; q++) {
      orc_topk_t t = {heap, 0, k};
      int nnz = sparsify(xq + (size_t)q * d, d, idx, val);
      for (int64_t i = 0; i < nb; i++)
        topk_push(&t, ip_sparse_q(idx, val, nnz, xb + (size_t)i * d), i);
      topk_finish(&t, D + (size_t)q * k, I + (size_t)q * k);
    }
    free(heap);
    free(idx);
    free(val);
  }
}

/* FAISS IndexRefineFlat restated: the ids of a short-list (any order, -1 = empty) are rescored
 * with the exact inner product against the stored vectors and the k best are returned,
 * (score desc, id asc), -1 / -FLT_MAX padded. */
void orc_refine(const float *xb, const float *xq, int32_t nq, int32_t d, const int64_t *I_in,
                int32_t kp, int32_t k, float *D, int64_t *I) {
#pragma omp parallel
  {
    orc_hit_t *heap = (orc_hit_t *)malloc(sizeof(orc_hit_t) * (size_t)(k > 0 ? k : 1));
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)d);
    float *val = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(dynamic, 4)
    for (int32_t q = 0; q < nq; q++) {
      orc_topk_t t = {heap, 0, k};
      int nnz = sparsify(xq + (size_t)q * d, d, idx, val);
      for (int32_t c = 0; c < kp; c++) {
        int64_t id = I_in[(size_t)q * kp + c];
        if (id >= 0) topk_push(&t, ip_sparse_q(idx, val, nnz, xb + (size_t)id * d), id);
      }
      topk_finish(&t, D + (size_t)q * k, I + (size_t)q * k);
    }
    free(heap);
    free(idx);
    free(val);
  }
}

/* ------------------------------------------------------------------------ */
/* k-means (FAISS Clustering restated: Lloyd, quantizer-metric assignment,   */
/* mean update, empty-cluster split with eps = 1/1024)                       */
/* ------------------------------------------------------------------------ */

static inline uint64_t sm64(uint64_t *s) {
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

void orc_rand_perm(int64_t n, uint64_t seed, int64_t *perm) {
  uint64_t s = seed;
  for (int64_t i = 0; i < n; i++) perm[i] = i;
  for (int64_t i = 0; i + 1 < n; i++) {
    int64_t j = i + (int64_t)(sm64(&s) % (uint64_t)(n - i));
    int64_t t = perm[i];
    perm[i] = perm[j];
    perm[j] = t;
  }
}

void orc_assign(const float *x, int64_t n, int32_t d, const float *centroids, int32_t k,
                int metric, int32_t *assign) {
#pragma omp parallel
  {
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)d);
    float *val = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; i++) {
      const float *xi = x + (size_t)i * d;
      int32_t best = 0;
      if (metric == ORC_METRIC_IP) {
        int nnz = sparsify(xi, d, idx, val);
        float bs = -INFINITY;
        for (int32_t c = 0; c < k; c++) {
          float s = ip_sparse_q(idx, val, nnz, centroids + (size_t)c * d);
          if (s > bs) {
            bs = s;
            best = c;
          }
        }
      } else {
        float bs = INFINITY;
        for (int32_t c = 0; c < k; c++) {
          float s = l2_chain(xi, centroids + (size_t)c * d, d);
          if (s < bs) {
            bs = s;
            best = c;
          }
        }
      }
      assign[i] = best;
    }
    free(idx);
    free(val);
  }
}

/* FAISS ClusteringParameters.spherical (post_process_centroids -> fvec_renorm_L2): every
 * centroid with a non-zero norm is rescaled to unit L2 length. Canonical arithmetic of this
 * build: ascending-index fmaf chain, sqrtf, IEEE divide -- the encoder's norm. */
static void renorm_rows(float *c, int32_t k, int32_t d) {
  for (int32_t i = 0; i < k; i++) {
    float *ci = c + (size_t)i * d;
    float acc = 0.0f;
    for (int32_t j = 0; j < d; j++) acc = fmaf(ci[j], ci[j], acc);
    if (acc > 0.0f) {
      float nrm = sqrtf(acc);
      for (int32_t j = 0; j < d; j++) ci[j] = ci[j] / nrm;
    }
  }
}

/* Lloyd k-means as FAISS' Clustering::train runs it (published algorithm; FAISS itself is
 * absent here): subsample to max_ppc points per centroid, k random training points as the
 * initial centroids, then niter x { assign by the quantiser's metric, mean update,
 * split_clusters for empty clusters (eps = 1/1024), post_process_centroids }.
 * metric == ORC_METRIC_IP implies SPHERICAL k-means: FAISS' IndexIVF constructor sets
 * cp.spherical = true for METRIC_INNER_PRODUCT (the reference builds exactly such an index,
 * src/ann_solo/spectral_library.py:174-178), so the centroids are L2-renormalised after the
 * initialisation and after every iteration. The PQ sub-quantisers (L2) are not spherical. */
void orc_kmeans(const float *x, int64_t n, int32_t d, int32_t k, int32_t niter,
                uint64_t seed, int metric, int32_t max_ppc, float *centroids) {
  const int spherical = metric == ORC_METRIC_IP;
  /* subsample to k*max_ppc points in permutation order */
  int64_t nt = n;
  const float *xt = x;
  float *sub = NULL;
  if (max_ppc > 0 && n > (int64_t)k * max_ppc) {
    nt = (int64_t)k * max_ppc;
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    orc_rand_perm(n, seed, perm);
    sub = (float *)malloc(sizeof(float) * (size_t)nt * d);
    for (int64_t i = 0; i < nt; i++)
      memcpy(sub + (size_t)i * d, x + (size_t)perm[i] * d, sizeof(float) * (size_t)d);
    free(perm);
    xt = sub;
  }
  /* init: k distinct random training points */
  {
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)nt);
    orc_rand_perm(nt, seed + 1, perm);
    for (int32_t c = 0; c < k; c++)
      memcpy(centroids + (size_t)c * d, xt + (size_t)perm[c % nt] * d,
             sizeof(float) * (size_t)d);
    free(perm);
  }
  if (spherical) renorm_rows(centroids, k, d);
  int32_t *assign = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt);
  float *hassign = (float *)malloc(sizeof(float) * (size_t)k);
  uint64_t rng = seed + 2;
  for (int it = 0; it < niter; it++) {
    orc_assign(xt, nt, d, centroids, k, metric, assign);
    /* mean update, ascending point order, fp32 */
    float *sum = (float *)calloc((size_t)k * d, sizeof(float));
    for (int32_t c = 0; c < k; c++) hassign[c] = 0.0f;
    for (int64_t i = 0; i < nt; i++) {
      int32_t c = assign[i];
      hassign[c] += 1.0f;
      float *sc = sum + (size_t)c * d;
      const float *xi = xt + (size_t)i * d;
      for (int32_t j = 0; j < d; j++) sc[j] += xi[j];
    }
    for (int32_t c = 0; c < k; c++)
      if (hassign[c] > 0.0f)
        for (int32_t j = 0; j < d; j++)
          centroids[(size_t)c * d + j] = sum[(size_t)c * d + j] / hassign[c];
    free(sum);
    /* empty-cluster split (FAISS Clustering.cpp split_clusters) */
    const float eps = 1.0f / 1024.0f;
    for (int32_t ci = 0; ci < k; ci++) {
      if (hassign[ci] != 0.0f) continue;
      int32_t cj = 0;
      for (int guard = 0; guard < 64 * k + 64; guard++, cj = (cj + 1) % k) {
        float p = (hassign[cj] - 1.0f) / (float)(nt - k);
        float r = (float)(sm64(&rng) >> 40) * (1.0f / 16777216.0f);
        if (r < p) break;
      }
      float *a = centroids + (size_t)ci * d, *b = centroids + (size_t)cj * d;
      memcpy(a, b, sizeof(float) * (size_t)d);
      for (int32_t j = 0; j < d; j++) {
        if (j % 2 == 0) {
          a[j] *= 1 + eps;
          b[j] *= 1 - eps;
        } else {
          a[j] *= 1 - eps;
          b[j] *= 1 + eps;
        }
      }
      hassign[ci] = floorf(hassign[cj] / 2);
      hassign[cj] -= hassign[ci];
    }
    if (spherical) renorm_rows(centroids, k, d);
  }
  free(assign);
  free(hassign);
  free(sub);
}

/* ------------------------------------------------------------------------ */
/* Product quantizer (by-residual, inner-product ADC)                        */
/* ------------------------------------------------------------------------ */

void orc_pq_train(const float *x, int64_t n, int32_t d, const float *centroids,
                  int32_t nlist, int32_t m, int32_t ksub, int32_t niter, uint64_t seed,
                  float *codebooks) {
  int32_t dsub = d / m;
  int64_t nt = n;
  int64_t cap = (int64_t)ksub * 256;
  int64_t *perm = NULL;
  if (nt > cap) {
    nt = cap;
    perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    orc_rand_perm(n, seed, perm);
  }
  float *xt = (float *)malloc(sizeof(float) * (size_t)nt * d);
  for (int64_t i = 0; i < nt; i++)
    memcpy(xt + (size_t)i * d, x + (size_t)(perm ? perm[i] : i) * d,
           sizeof(float) * (size_t)d);
  free(perm);
  int32_t *assign = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt);
  orc_assign(xt, nt, d, centroids, nlist, ORC_METRIC_IP, assign);
  for (int64_t i = 0; i < nt; i++) {
    const float *c = centroids + (size_t)assign[i] * d;
    float *r = xt + (size_t)i * d;
    for (int32_t j = 0; j < d; j++) r[j] = r[j] - c[j];
  }
  free(assign);
  float *sub = (float *)malloc(sizeof(float) * (size_t)nt * dsub);
  for (int32_t mi = 0; mi < m; mi++) {
    for (int64_t i = 0; i < nt; i++)
      memcpy(sub + (size_t)i * dsub, xt + (size_t)i * d + (size_t)mi * dsub,
             sizeof(float) * (size_t)dsub);
    orc_kmeans(sub, nt, dsub, ksub, niter, seed + 16 + (uint64_t)mi, ORC_METRIC_L2, 0,
               codebooks + (size_t)mi * ksub * dsub);
  }
  free(sub);
  free(xt);
}

void orc_pq_encode(const float *x, int64_t n, int32_t d, const float *centroids,
                   const int32_t *assign, const float *codebooks, int32_t m,
                   int32_t ksub, uint8_t *codes) {
  int32_t dsub = d / m;
#pragma omp parallel
  {
    float *r = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; i++) {
      const float *c = centroids + (size_t)assign[i] * d;
      const float *xi = x + (size_t)i * d;
      for (int32_t j = 0; j < d; j++) r[j] = xi[j] - c[j];
      for (int32_t mi = 0; mi < m; mi++) {
        const float *cb = codebooks + (size_t)mi * ksub * dsub;
        float bs = INFINITY;
        int32_t best = 0;
        for (int32_t e = 0; e < ksub; e++) {
          float s = l2_chain(r + (size_t)mi * dsub, cb + (size_t)e * dsub, dsub);
          if (s < bs) {
            bs = s;
            best = e;
          }
        }
        codes[(size_t)i * m + mi] = (uint8_t)best;
      }
    }
    free(r);
  }
}

void orc_pq_lut(const float *xq, int32_t d, const float *codebooks, int32_t m,
                int32_t ksub, float *lut) {
  int32_t dsub = d / m;
  for (int32_t mi = 0; mi < m; mi++)
    for (int32_t e = 0; e < ksub; e++)
      lut[(size_t)mi * ksub + e] =
          orc_ip(xq + (size_t)mi * dsub, codebooks + ((size_t)mi * ksub + e) * dsub, dsub);
}

float orc_adc(const float *lut, int32_t m, int32_t ksub, const uint8_t *code, float coarse) {
  float p[16];
  for (int j = 0; j < 16; j++) {
    float a = 0.0f;
    int first = 1;
    for (int mi = j; mi < m; mi += 16) {
      float v = lut[(size_t)mi * ksub + code[mi]];
      a = first ? v : a + v;
      first = 0;
    }
    p[j] = a;
  }
  /* fixed "mirror" tree: (j,15-j) (j,7-j) (j,3-j) (0,1) */
  for (int j = 0; j < 8; j++) p[j] = p[j] + p[15 - j];
  for (int j = 0; j < 4; j++) p[j] = p[j] + p[7 - j];
  for (int j = 0; j < 2; j++) p[j] = p[j] + p[3 - j];
  return coarse + (p[0] + p[1]);
}

/* ------------------------------------------------------------------------ */
/* IVF search                                                                */
/* ------------------------------------------------------------------------ */

static void coarse_one(const float *q, int32_t d, const float *centroids, int32_t nlist,
                       int32_t nprobe, orc_hit_t *heap, int32_t *idx, float *val,
                       float *cD, int32_t *cI) {
  orc_topk_t t = {heap, 0, nprobe};
  int nnz = sparsify(q, d, idx, val);
  for (int32_t c = 0; c < nlist; c++)
    topk_push(&t, ip_sparse_q(idx, val, nnz, centroids + (size_t)c * d), c);
  qsort(t.h, (size_t)t.n, sizeof(orc_hit_t), hit_cmp);
  for (int32_t i = 0; i < nprobe; i++) {
    cD[i] = i < t.n ? t.h[i].s : -FLT_MAX;
    cI[i] = i < t.n ? (int32_t)t.h[i].id : -1;
  }
}

void orc_coarse(const float *xq, int32_t nq, int32_t d, const float *centroids,
                int32_t nlist, int32_t nprobe, float *coarse_D, int32_t *coarse_I) {
#pragma omp parallel
  {
    orc_hit_t *heap = (orc_hit_t *)malloc(sizeof(orc_hit_t) * (size_t)nprobe);
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)d);
    float *val = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(static)
    for (int32_t q = 0; q < nq; q++)
      coarse_one(xq + (size_t)q * d, d, centroids, nlist, nprobe, heap, idx, val,
                 coarse_D + (size_t)q * nprobe, coarse_I + (size_t)q * nprobe);
    free(heap);
    free(idx);
    free(val);
  }
}

typedef struct {
  orc_hit_t *heap, *cheap;
  int32_t *idx, *cI;
  float *val, *cD, *lut;
} ivf_ws_t;

static void ivf_ws_alloc(ivf_ws_t *w, int32_t d, int32_t k, int32_t nprobe, int32_t m,
                         int32_t ksub) {
  w->heap = (orc_hit_t *)malloc(sizeof(orc_hit_t) * (size_t)(k > 0 ? k : 1));
  w->cheap = (orc_hit_t *)malloc(sizeof(orc_hit_t) * (size_t)nprobe);
  w->idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)d);
  w->val = (float *)malloc(sizeof(float) * (size_t)d);
  w->cD = (float *)malloc(sizeof(float) * (size_t)nprobe);
  w->cI = (int32_t *)malloc(sizeof(int32_t) * (size_t)nprobe);
  w->lut = m > 0 ? (float *)malloc(sizeof(float) * (size_t)m * ksub) : NULL;
}
static void ivf_ws_free(ivf_ws_t *w) {
  free(w->heap);
  free(w->cheap);
  free(w->idx);
  free(w->val);
  free(w->cD);
  free(w->cI);
  free(w->lut);
}

/* IndexIVFFlat.search (spectral_library.py:443-444): exact IP over the
 * vectors of the nprobe lists with the largest centroid IP. */
static void ivfflat_one(const float *q, int32_t d, const float *centroids, int32_t nlist,
                        const int32_t *list_offsets, const int32_t *ids, const float *vecs,
                        int32_t k, int32_t nprobe, ivf_ws_t *w, float *D, int64_t *I) {
  coarse_one(q, d, centroids, nlist, nprobe, w->cheap, w->idx, w->val, w->cD, w->cI);
  orc_topk_t t = {w->heap, 0, k};
  int nnz = sparsify(q, d, w->idx, w->val);
  for (int32_t p = 0; p < nprobe; p++) {
    int32_t l = w->cI[p];
    if (l < 0) continue;
    for (int32_t i = list_offsets[l]; i < list_offsets[l + 1]; i++)
      topk_push(&t, ip_sparse_q(w->idx, w->val, nnz, vecs + (size_t)i * d), ids[i]);
  }
  topk_finish(&t, D, I);
}

void orc_ivfflat_search(const float *xq, int32_t nq, int32_t d, const float *centroids,
                        int32_t nlist, const int32_t *list_offsets, const int32_t *ids,
                        const float *vecs, int32_t k, int32_t nprobe, float *D, int64_t *I) {
  if (nprobe > nlist) nprobe = nlist;
#pragma omp parallel
  {
    ivf_ws_t w;
    ivf_ws_alloc(&w, d, k, nprobe, 0, 0);
#pragma omp for schedule(dynamic, 4)
    for (int32_t q = 0; q < nq; q++)
      ivfflat_one(xq + (size_t)q * d, d, centroids, nlist, list_offsets, ids, vecs, k,
                  nprobe, &w, D + (size_t)q * k, I + (size_t)q * k);
    ivf_ws_free(&w);
  }
}

/* Fixed-point storage of IVF-Flat components (asl_index_set_flat_storage, ASL_FLAT_FX22; no
 * FAISS CPU counterpart -- the reference's GPU clone stores float16, spectral_library.py:490-497):
 * a component in [0, 1) is stored as the nearest multiple of 2^-22 (ties to even, at most
 * 1 - 2^-22), anything else as given. add() applies it; the scores are the same ascending fmaf
 * chain over the STORED components. */
float orc_fx22(float x) {
  if (!(x >= 0.0f && x < 1.0f)) return x;
  float m = rintf(x * 4194304.0f);
  return (m < 4194303.0f ? m : 4194303.0f) * (1.0f / 4194304.0f);
}
void orc_quantize_fx22(float *x, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) x[i] = orc_fx22(x[i]);
}

/* ---- the same IVF-Flat search over a SPARSE copy of the stored vectors (CSR in list order:
 * row i = the non-zeros of the vector at list position i, dimensions ascending). A hashed
 * spectrum has <= ~50 non-zeros of 800: the score of a stored vector is the ascending chain
 * acc = fmaf(q[dim], val, acc) over ITS non-zeros -- a zero query component leaves the
 * accumulator unchanged bit for bit (the accumulator is never -0), so this equals
 * ivfflat_one's chain over the query's non-zeros, which equals the dense chain. What a
 * sparse-aware CPU implementation does: 6 bytes per stored non-zero instead of a 3 200-byte
 * row per scanned vector (bench.py: the CPU baseline of the IVF-Flat leg). */
int64_t orc_csr_count(const float *x, int64_t n, int32_t d) {
  int64_t c = 0;
#pragma omp parallel for schedule(static) reduction(+ : c)
  for (int64_t i = 0; i < n * (int64_t)d; i++) c += x[i] != 0.0f;
  return c;
}
void orc_csr_fill(const float *x, int64_t n, int32_t d, int64_t *indptr, uint16_t *dims,
                  float *vals) {
  indptr[0] = 0;
  for (int64_t i = 0; i < n; i++) {
    int64_t c = 0;
    for (int32_t j = 0; j < d; j++) c += x[(size_t)i * d + j] != 0.0f;
    indptr[i + 1] = indptr[i] + c;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    int64_t o = indptr[i];
    for (int32_t j = 0; j < d; j++)
      if (x[(size_t)i * d + j] != 0.0f) {
        dims[o] = (uint16_t)j;
        vals[o] = x[(size_t)i * d + j];
        o++;
      }
  }
}

static void ivfflat_csr_one(const float *q, int32_t d, const float *centroids, int32_t nlist,
                            const int32_t *list_offsets, const int32_t *ids,
                            const orc_csr_t *csr, int32_t k, int32_t nprobe, ivf_ws_t *w,
                            float *D, int64_t *I) {
  coarse_one(q, d, centroids, nlist, nprobe, w->cheap, w->idx, w->val, w->cD, w->cI);
  orc_topk_t t = {w->heap, 0, k};
  for (int32_t p = 0; p < nprobe; p++) {
    int32_t l = w->cI[p];
    if (l < 0) continue;
    for (int32_t i = list_offsets[l]; i < list_offsets[l + 1]; i++) {
      float acc = 0.0f;
      for (int64_t o = csr->indptr[i]; o < csr->indptr[i + 1]; o++)
        acc = fmaf(q[csr->dims[o]], csr->vals[o], acc);
      topk_push(&t, acc, ids[i]);
    }
  }
  topk_finish(&t, D, I);
}

void orc_ivfflat_csr_search(const float *xq, int32_t nq, int32_t d, const float *centroids,
                            int32_t nlist, const int32_t *list_offsets, const int32_t *ids,
                            const int64_t *indptr, const uint16_t *dims, const float *vals,
                            int32_t k, int32_t nprobe, float *D, int64_t *I) {
  if (nprobe > nlist) nprobe = nlist;
  orc_csr_t csr = {indptr, dims, vals};
#pragma omp parallel
  {
    ivf_ws_t w;
    ivf_ws_alloc(&w, d, k, nprobe, 0, 0);
#pragma omp for schedule(dynamic, 4)
    for (int32_t q = 0; q < nq; q++)
      ivfflat_csr_one(xq + (size_t)q * d, d, centroids, nlist, list_offsets, ids, &csr, k,
                      nprobe, &w, D + (size_t)q * k, I + (size_t)q * k);
    ivf_ws_free(&w);
  }
}

static void ivfpq_one(const float *q, int32_t d, const float *centroids, int32_t nlist,
                      const int32_t *list_offsets, const int32_t *ids, const uint8_t *codes,
                      const float *codebooks, int32_t m, int32_t ksub, int32_t k,
                      int32_t nprobe, ivf_ws_t *w, float *D, int64_t *I) {
  coarse_one(q, d, centroids, nlist, nprobe, w->cheap, w->idx, w->val, w->cD, w->cI);
  orc_pq_lut(q, d, codebooks, m, ksub, w->lut);
  orc_topk_t t = {w->heap, 0, k};
  for (int32_t p = 0; p < nprobe; p++) {
    int32_t l = w->cI[p];
    if (l < 0) continue;
    float coarse = w->cD[p];
    for (int32_t i = list_offsets[l]; i < list_offsets[l + 1]; i++)
      topk_push(&t, orc_adc(w->lut, m, ksub, codes + (size_t)i * m, coarse), ids[i]);
  }
  topk_finish(&t, D, I);
}

void orc_ivfpq_search(const float *xq, int32_t nq, int32_t d, const float *centroids,
                      int32_t nlist, const int32_t *list_offsets, const int32_t *ids,
                      const uint8_t *codes, const float *codebooks, int32_t m,
                      int32_t ksub, int32_t k, int32_t nprobe, float *D, int64_t *I) {
  if (nprobe > nlist) nprobe = nlist;
#pragma omp parallel
  {
    ivf_ws_t w;
    ivf_ws_alloc(&w, d, k, nprobe, m, ksub);
#pragma omp for schedule(dynamic, 4)
    for (int32_t q = 0; q < nq; q++)
      ivfpq_one(xq + (size_t)q * d, d, centroids, nlist, list_offsets, ids, codes,
                codebooks, m, ksub, k, nprobe, &w, D + (size_t)q * k, I + (size_t)q * k);
    ivf_ws_free(&w);
  }
}

void orc_topk_merge(const float *Ds, const int64_t *Is, int32_t S, int32_t nq, int32_t k,
                    float *D, int64_t *I) {
#pragma omp parallel
  {
    orc_hit_t *heap = (orc_hit_t *)malloc(sizeof(orc_hit_t) * (size_t)(k > 0 ? k : 1));
#pragma omp for schedule(static)
    for (int32_t q = 0; q < nq; q++) {
      orc_topk_t t = {heap, 0, k};
      for (int32_t s = 0; s < S; s++)
        for (int32_t i = 0; i < k; i++) {
          size_t o = ((size_t)s * nq + q) * k + i;
          if (Is[o] >= 0) topk_push(&t, Ds[o], Is[o]);
        }
      topk_finish(&t, D + (size_t)q * k, I + (size_t)q * k);
    }
    free(heap);
  }
}

/* ------------------------------------------------------------------------ */
/* Precursor window: spectral_library.py:417-429 (numexpr promotes to f64)   */
/* ------------------------------------------------------------------------ */

int orc_precursor_ok(double q_mz, float lib_mz, int32_t charge, double tol, int mode) {
  double l = (double)lib_mz;
  if (mode == ORC_TOL_DA) return fabs(q_mz - l) * (double)charge <= tol;
  return fabs(q_mz - l) / l * 1000000.0 <= tol;
}

/* ------------------------------------------------------------------------ */
/* One batch of the hot path: spectral_library.py:328-455                    */
/* ------------------------------------------------------------------------ */

static int cmp_i64(const void *a, const void *b) {
  int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return x < y ? -1 : x > y;
}

void orc_search_batch(const orc_peaks_t *Q, const orc_peaks_t *L, const float *lib_pmz_f32,
                      int32_t charge, double min_bound, double bin_size, int32_t d,
                      uint32_t seed, int kind, const float *centroids, int32_t nlist,
                      const int32_t *list_offsets, const int32_t *ids, const void *payload,
                      const float *codebooks, int32_t m, int32_t ksub, int32_t k,
                      int32_t nprobe, double prec_tol, int prec_mode, double frag_tol,
                      int allow_shift, int32_t *best_row, double *best_score,
                      int32_t *n_cand, int32_t *pm_count, uint32_t *pm_pairs,
                      int32_t pm_stride, int64_t *knn_I, int32_t nthreads) {
  if (nprobe > nlist) nprobe = nlist;
#ifdef _OPENMP
  int prev = omp_get_max_threads();
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    ivf_ws_t w;
    ivf_ws_alloc(&w, d, k, nprobe, kind == 1 ? m : 0, ksub);
    float *vec = (float *)malloc(sizeof(float) * (size_t)d);
    float *Dk = (float *)malloc(sizeof(float) * (size_t)k);
    int64_t *Ik = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)k);
    uint32_t *pm = (uint32_t *)malloc(sizeof(uint32_t) * 2 * 4096);
#pragma omp for schedule(dynamic, 4)
    for (int32_t q = 0; q < Q->n; q++) {
      int32_t qo = Q->offsets[q], qn = Q->offsets[q + 1] - qo;
      /* spectral_library.py:435-440 */
      orc_encode(Q->mz + qo, Q->intensity + qo, qn, min_bound, bin_size, d, seed, 1, vec);
      /* :443-444 */
      if (kind == 1)
        ivfpq_one(vec, d, centroids, nlist, list_offsets, ids, (const uint8_t *)payload,
                  codebooks, m, ksub, k, nprobe, &w, Dk, Ik);
      else if (kind == 2)
        ivfflat_csr_one(vec, d, centroids, nlist, list_offsets, ids, (const orc_csr_t *)payload,
                        k, nprobe, &w, Dk, Ik);
      else
        ivfflat_one(vec, d, centroids, nlist, list_offsets, ids, (const float *)payload, k,
                    nprobe, &w, Dk, Ik);
      if (knn_I) memcpy(knn_I + (size_t)q * k, Ik, sizeof(int64_t) * (size_t)k);
      /* :417-429 AND :441-446 (post-filter), candidates in ascending row order (:451) */
      int nc = 0;
      for (int32_t i = 0; i < k; i++)
        if (Ik[i] >= 0 &&
            orc_precursor_ok(Q->precursor_mz[q], lib_pmz_f32[Ik[i]], charge, prec_tol,
                             prec_mode))
          cand[nc++] = Ik[i];
      qsort(cand, (size_t)nc, sizeof(int64_t), cmp_i64);
      n_cand[q] = nc;
      /* :356-365 */
      int32_t nm = 0;
      double sc = 0.0;
      int32_t b = orc_best_match(Q, q, L, cand, nc, frag_tol, allow_shift, &sc,
                                 pm_pairs ? pm : NULL, &nm);
      best_row[q] = b >= 0 ? (int32_t)cand[b] : -1;
      best_score[q] = b >= 0 ? sc : 0.0;
      if (pm_count) pm_count[q] = nm;
      if (pm_pairs) {
        int32_t w2 = nm < pm_stride ? nm : pm_stride;
        memcpy(pm_pairs + (size_t)q * pm_stride * 2, pm, sizeof(uint32_t) * 2 * (size_t)w2);
      }
    }
    ivf_ws_free(&w);
    free(vec);
    free(Dk);
    free(Ik);
    free(cand);
    free(pm);
  }
#ifdef _OPENMP
  omp_set_num_threads(prev);
#endif
}

// encode.hip -- feature-hash encoder (replaces spectrum_to_vector,
// /root/reference/src/ann_solo/spectrum.py:146-214).
//
// One 64-lane wavefront per spectrum, four spectra per workgroup. Lanes hash
// their peaks in parallel (float64 NumPy floor-division of the float32 m/z,
// MurmurHash3_x86_32 of the decimal string of the bin index, seed 42); the fp32
// scatter-add into the LDS-resident vector is applied per bin in ascending
// peak order (rounds by a peak's rank within its bin), so that colliding bins
// round exactly like the reference's `vector[bin] += intensity` loop. The L2
// norm is the canonical ascending-index fmaf chain. Output rows are written
// coalesced, or as the entry lists the scans read (non-zero components only).
#include "common.hpp"

namespace asl {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) {
  return (x << r) | (x >> (32 - r));
}

__device__ __forceinline__ uint32_t murmur3_round(uint32_t h1, uint32_t k1) {
  k1 *= 0xcc9e2d51u;
  k1 = rotl32(k1, 15);
  k1 *= 0x1b873593u;
  h1 ^= k1;
  h1 = rotl32(h1, 13);
  return h1 * 5 + 0xe6546b64u;
}

// MurmurHash3_x86_32 over the ASCII decimal representation of v (with '-'), |v| < 10^9 (every
// bin of a real spectrum: m/z / bin size): at most 10 characters, kept in three 32-bit words --
// 32-bit digit arithmetic and no byte arrays (the general form below runs three data-dependent
// loops over 64-bit divisions and register-resident arrays: ~3 000 instructions per peak, which
// made the encoder VALU-bound). Same bytes, same hash.
__device__ __forceinline__ uint32_t murmur3_decimal_small(int v, uint32_t seed) {
  const bool neg = v < 0;
  uint32_t u = neg ? (uint32_t)(-(long long)v) : (uint32_t)v;
  int nd = 1;
  nd += u >= 10u;
  nd += u >= 100u;
  nd += u >= 1000u;
  nd += u >= 10000u;
  nd += u >= 100000u;
  nd += u >= 1000000u;
  nd += u >= 10000000u;
  nd += u >= 100000000u;
  const int len = nd + (neg ? 1 : 0);
  uint32_t w0 = neg ? (uint32_t)'-' : 0u, w1 = 0u, w2 = 0u;      // byte p of the string = byte p & 3 of word p >> 2
#pragma unroll
  for (int j = 0; j < 9; ++j) {          // digit j from the right goes to position len - 1 - j
    if (j < nd) {
      const uint32_t qd = u / 10u;
      const uint32_t ch = (uint32_t)'0' + (u - qd * 10u);
      u = qd;
      const int pos = len - 1 - j;
      const uint32_t sh = ch << ((pos & 3) * 8);
      w0 |= (pos >> 2) == 0 ? sh : 0u;
      w1 |= (pos >> 2) == 1 ? sh : 0u;
      w2 |= (pos >> 2) == 2 ? sh : 0u;
    }
  }
  uint32_t h1 = seed;
  if (len >= 4) h1 = murmur3_round(h1, w0);
  if (len >= 8) h1 = murmur3_round(h1, w1);
  const uint32_t tail = len >= 8 ? w2 : (len >= 4 ? w1 : w0);    // the len & 3 bytes behind the full blocks
  if (len & 3) {
    uint32_t k1 = tail & (0xFFFFFFFFu >> (32 - 8 * (len & 3)));
    k1 *= 0xcc9e2d51u;
    k1 = rotl32(k1, 15);
    k1 *= 0x1b873593u;
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

// MurmurHash3_x86_32 over the ASCII decimal representation of v (with '-').
__device__ uint32_t murmur3_decimal(long long v, uint32_t seed) {
  if (v > -1000000000ll && v < 1000000000ll) return murmur3_decimal_small((int)v, seed);
  unsigned char buf[24];
  int len = 0;
  unsigned long long u = v < 0 ? (unsigned long long)(-(v + 1)) + 1ull : (unsigned long long)v;
  unsigned char tmp[24];
  int nd = 0;
  do {
    tmp[nd++] = (unsigned char)('0' + (u % 10));
    u /= 10;
  } while (u);
  if (v < 0) buf[len++] = '-';
  for (int i = nd - 1; i >= 0; --i) buf[len++] = tmp[i];
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  uint32_t h1 = seed;
  int nblocks = len >> 2;
  for (int i = 0; i < nblocks; ++i) {
    uint32_t k1 = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) |
                  ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
    k1 *= c1;
    k1 = rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
  }
  uint32_t k1 = 0;
  int t = nblocks * 4;
  int rem = len & 3;
  if (rem == 3) k1 ^= (uint32_t)buf[t + 2] << 16;
  if (rem >= 2) k1 ^= (uint32_t)buf[t + 1] << 8;
  if (rem >= 1) {
    k1 ^= buf[t];
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

// NumPy npy_floor_divide on doubles, then math.floor (spectrum.py:207).
__device__ long long np_floor_div_bin(double a, double b) {
  // NumPy's route (exact remainder by fmod, quotient of the difference, the corrections below) yields
  // the EXACT floor of a / b for quotients far inside the double range; so does this one, without
  // the fmod loop: the correctly rounded quotient can only have been rounded UP across an integer
  // (an integer n <= a / b is representable, rounding is monotone), and the sign of the fused
  // a - q b is the sign of the exact remainder. b > 0 (checked by the callers); everything else --
  // huge quotients, infinities, NaN -- takes NumPy's route.
  const double q0 = a / b;
  if (b > 0.0 && fabs(q0) < 2147483648.0) {
    double q = floor(q0);
    if (fma(-q, b, a) < 0.0) q -= 1.0;
    return (long long)q;
  }
  double mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.0 && ((b < 0) != (mod < 0))) div -= 1.0;
  double fd;
  if (div != 0.0) {
    fd = floor(div);
    if (div - fd > 0.5) fd += 1.0;
  } else {
    fd = copysign(0.0, a / b);
  }
  return (long long)floor(fd);
}

constexpr int ENC_WAVES = 4;
constexpr int QE_CAP = 64;         // entries of a query's entry list (coarse_sparse.hip: CS_CAP)
constexpr int QE_DIM_SHIFT = 7;    // an entry holds dimension * 128: the byte offset of its tile row (CS_TL * 4)

// ENTRIES = false: out[spec][hash_len], the dense vector. ENTRIES = true: the vector leaves as its
// ENTRY LIST -- the non-zero components, ascending, as (dimension * 128, value bits), 64 per query,
// cnt = their number or -1 - count beyond 64 -- exactly what list_nonzeros_kernel
// (coarse_sparse.hip) makes of the dense row; the scans take their queries in this form, and a
// sharded search encodes the other ranks' queries straight into it (no 3.2 KB row per query
// written and read again).
template <bool ENTRIES>
__global__ __launch_bounds__(64 * ENC_WAVES) void encode_kernel(
    const float *__restrict__ mz, const float *__restrict__ inten,
    const int32_t *__restrict__ offsets, int32_t n, double min_bound, double bin_size,
    int32_t hash_len, uint32_t seed, int norm, float *__restrict__ out, uint2 *__restrict__ ent,
    int32_t *__restrict__ ent_cnt, int *__restrict__ n_over) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int spec = blockIdx.x * ENC_WAVES + wave;
  float *vec = reinterpret_cast<float *>(smem) + (size_t)wave * hash_len;   // per-wave LDS: vec[hash_len]
  if (spec >= n) return;  // whole wave exits together (no block barriers are used)

  for (int i = lane; i < hash_len; i += 64) vec[i] = 0.0f;
  const int p0 = offsets[spec], p1 = offsets[spec + 1];
  for (int base = p0; base < p1; base += 64) {
    const int p = base + lane;
    const int cnt = min(64, p1 - base);
    const bool live = lane < cnt;
    int idx = -1 - lane;        // (a dead lane matches nobody)
    float val = 0.0f;
    if (live) {
      long long b = np_floor_div_bin((double)mz[p] - min_bound, bin_size);
      idx = (int)(murmur3_decimal(b, seed) % (uint32_t)hash_len);
      val = inten[p];
    }
    // `vector[bin] += intensity` in ascending peak order: peaks of different bins do not interact, so
    // a lane's turn is its RANK among the peaks of its own bin (the earlier lanes holding the same
    // bin) -- round r applies all peaks of rank r at once, one read-modify-write per bin and round.
    // Two or three rounds instead of 50 dependent LDS round trips of one lane.
    int rank = 0;
    for (int j = 0; j < cnt; ++j) {                 // wave-uniform trip count
      const int o = __builtin_amdgcn_readlane(idx, j);
      rank += (o == idx && j < lane) ? 1 : 0;
    }
    for (int r = 0; __ballot(live && rank >= r) != 0ull; ++r) {
      if (live && rank == r) vec[idx] += val;
      __builtin_amdgcn_wave_barrier();              // (a wave's LDS accesses execute in program order)
    }
  }
  float nrm = 1.0f;
  if (norm) {
    // ascending-index fmaf chain over the non-zero components (a zero leaves the accumulator
    // unchanged): the wave finds them 64 at a time with a ballot and every lane walks the same
    // chain on broadcast values -- ~50 steps instead of hash_len serial LDS reads by one lane
    float acc = 0.0f;
    for (int i0 = 0; i0 < hash_len; i0 += 64) {
      const int i = i0 + lane;
      const float v = i < hash_len ? vec[i] : 0.0f;
      unsigned long long m = __ballot(v != 0.0f);
      while (m) {                       // wave-uniform
        const int l = __builtin_ctzll(m);
        m &= m - 1ull;
        const float x = __builtin_bit_cast(
            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
        acc = __builtin_fmaf(x, x, acc);
      }
    }
    nrm = __builtin_sqrtf(acc);
  }
  if (!ENTRIES) {
    float *row = out + (size_t)spec * hash_len;
    for (int i = lane; i < hash_len; i += 64) row[i] = norm ? vec[i] / nrm : vec[i];
    return;
  }
  // the entry list of the row the dense form would hold (the test is on the STORED value, as
  // list_nonzeros applies it: a component that underflows to zero in the division is no entry)
  uint2 *row = ent + (size_t)spec * QE_CAP;
  int have = 0;
  for (int i0 = 0; i0 < hash_len; i0 += 64) {
    const int i = i0 + lane;
    float x = i < hash_len ? vec[i] : 0.0f;
    if (norm) x = x / nrm;
    if (!(i < hash_len)) x = 0.0f;
    const unsigned long long m = __ballot(x != 0.0f);
    if (x != 0.0f) {
      const int t = have + __popcll(m & ((1ull << lane) - 1ull));
      if (t < QE_CAP) row[t] = make_uint2((uint32_t)i << QE_DIM_SHIFT, __float_as_uint(x));
    }
    have += __popcll(m);
  }
  for (int t = have + lane; t < QE_CAP; t += 64) row[t] = make_uint2(0u, 0u);     // (dim 0, +0.0)
  if (lane == 0) {
    ent_cnt[spec] = have <= QE_CAP ? have : -1 - have;
    if (have > QE_CAP && n_over) atomicAdd(n_over, 1);
  }
}

template <bool ENTRIES>
static int encode_launch(const float *mz, const float *inten, const int32_t *offsets, int32_t n,
                         double min_bound, double bin_size, int32_t hash_len, uint32_t seed, int norm,
                         float *out, uint2 *ent, int32_t *ent_cnt, int *n_over) {
  size_t lds = (size_t)ENC_WAVES * hash_len * sizeof(float);
  if (lds > 160 * 1024) return fail(ASL_ERR_CAPACITY, "encode: hash_len %d too large for LDS", hash_len);
  dim3 grid((unsigned)cdiv(n, ENC_WAVES));
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)encode_kernel<ENTRIES>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(encode_kernel<ENTRIES>, grid, dim3(64 * ENC_WAVES), lds, stream(), mz, inten,
                     offsets, n, min_bound, bin_size, hash_len, seed, norm, out, ent, ent_cnt, n_over);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
}

int encode_device(const float *mz, const float *inten, const int32_t *offsets, int32_t n,
                  double min_bound, double bin_size, int32_t hash_len, uint32_t seed,
                  int norm, float *out) {
  if (n == 0) return ASL_OK;
  ProfScope ps("encode");
  return encode_launch<false>(mz, inten, offsets, n, min_bound, bin_size, hash_len, seed, norm, out,
                              nullptr, nullptr, nullptr);
}

// the same vectors as entry lists: ent [n][64] (dimension * 128, value bits), cnt [n], n_over: one
// device int that counts the rows with more than 64 non-zeros (may be null)
int encode_entries_device(const float *mz, const float *inten, const int32_t *offsets, int32_t n,
                          double min_bound, double bin_size, int32_t hash_len, uint32_t seed, int norm,
                          uint2 *ent, int32_t *cnt, int *n_over) {
  if (n == 0) return ASL_OK;
  ProfScope ps("encode");
  return encode_launch<true>(mz, inten, offsets, n, min_bound, bin_size, hash_len, seed, norm, nullptr,
                             ent, cnt, n_over);
}

}  // namespace asl

using namespace asl;

// ---- host-side integer helpers (get_dim / hash_idx are pure scalar functions) ----
static uint32_t host_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint32_t host_murmur3(const unsigned char *key, int len, uint32_t seed) {
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  uint32_t h1 = seed;
  int nblocks = len / 4;
  for (int i = 0; i < nblocks; i++) {
    uint32_t k1;
    memcpy(&k1, key + 4 * i, 4);
    k1 *= c1;
    k1 = host_rotl32(k1, 15);
    k1 *= c2;
    h1 ^= k1;
    h1 = host_rotl32(h1, 13);
    h1 = h1 * 5 + 0xe6546b64u;
  }
  const unsigned char *tail = key + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; [[fallthrough]];
    case 2: k1 ^= (uint32_t)tail[1] << 8; [[fallthrough]];
    case 1:
      k1 ^= tail[0];
      k1 *= c1;
      k1 = host_rotl32(k1, 15);
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

extern "C" {

int32_t asl_hash_idx(int64_t bin_idx, int32_t hash_len, uint32_t seed) {
  if (hash_len <= 0) return -1;
  char buf[32];
  int len = snprintf(buf, sizeof buf, "%lld", (long long)bin_idx);
  return (int32_t)(host_murmur3((const unsigned char *)buf, len, seed) % (uint32_t)hash_len);
}

int asl_get_dim(double min_mz, double max_mz, double bin_size, int64_t *n_bins,
                double *start_dim, double *end_dim) {
  if (!(bin_size > 0)) return fail(ASL_ERR_INVALID, "get_dim: bin_size must be > 0");
  auto pymod = [](double a, double b) {
    double m = fmod(a, b);
    if (m != 0.0 && ((b < 0) != (m < 0))) m += b;
    return m;
  };
  double s = min_mz - pymod(min_mz, bin_size);
  double e = max_mz + bin_size - pymod(max_mz, bin_size);
  if (start_dim) *start_dim = s;
  if (end_dim) *end_dim = e;
  if (n_bins) *n_bins = (int64_t)nearbyint((e - s) / bin_size);
  return ASL_OK;
}

int asl_encode_batch(const float *mz, const float *intensity, const int32_t *offsets,
                     int32_t n, double min_bound, double bin_size, int32_t hash_len,
                     uint32_t seed, int norm, float *out) {
  clear_error();
  if (n < 0 || hash_len <= 0 || !(bin_size > 0))
    return fail(ASL_ERR_INVALID, "encode_batch: bad n/hash_len/bin_size");
  if (n == 0) return ASL_OK;
  if (!offsets || !out) return fail(ASL_ERR_INVALID, "encode_batch: null offsets/out");
  ASL_TRY(ensure_device());
  int32_t last = 0;
  if (is_device_ptr(offsets)) {
    HIP_TRY(hipMemcpyAsync(&last, offsets + n, sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
    ASL_TRY(sync_stream());
  } else {
    last = offsets[n];
  }
  if (last > 0 && (!mz || !intensity)) return fail(ASL_ERR_INVALID, "encode_batch: null peaks");
  In<float> dmz, din;
  In<int32_t> doff;
  Out<float> dout;
  ASL_TRY(dmz.init(mz, (size_t)last));
  ASL_TRY(din.init(intensity, (size_t)last));
  ASL_TRY(doff.init(offsets, (size_t)n + 1));
  ASL_TRY(dout.init(out, (size_t)n * hash_len));
  ASL_TRY(encode_device(dmz.d, din.d, doff.d, n, min_bound, bin_size, hash_len, seed, norm,
                        dout.d));
  ASL_TRY(dout.finish());
  if (dout.to_host() || dmz.own.p || din.own.p || doff.own.p) ASL_TRY(sync_stream());
  return ASL_OK;
}

// The same vectors as ENTRY LISTS, the form the IVF scans read their queries in
// (asl_index_search_entries): entries [n][64] pairs of 32-bit words (dimension * 128, the bits of
// the fp32 value), ascending dimension, unused pairs zero; counts [n] = the number of non-zero
// components, or -1 - count when a vector has more than 64 (such a query needs the dense form);
// n_over (may be null): a device int this call ADDS the number of such rows to. entries / counts
// are device memory; n_peaks = offsets[n] when the caller knows it (>= 0: nothing is read back).
int asl_encode_entries_batch(const float *mz, const float *intensity, const int32_t *offsets,
                             int32_t n, int32_t n_peaks, double min_bound, double bin_size,
                             int32_t hash_len, uint32_t seed, int norm, uint32_t *entries,
                             int32_t *counts, int32_t *n_over) {
  clear_error();
  if (n < 0 || hash_len <= 0 || !(bin_size > 0))
    return fail(ASL_ERR_INVALID, "encode_entries_batch: bad n/hash_len/bin_size");
  if (n == 0) return ASL_OK;
  if (!offsets || !entries || !counts) return fail(ASL_ERR_INVALID, "encode_entries_batch: null offsets/entries/counts");
  ASL_TRY(ensure_device());
  if (!is_device_ptr(entries) || !is_device_ptr(counts) || (n_over && !is_device_ptr(n_over)))
    return fail(ASL_ERR_INVALID, "encode_entries_batch: entries / counts / n_over must be device memory");
  int32_t last = n_peaks;
  if (last < 0) {
    if (is_device_ptr(offsets)) {
      HIP_TRY(hipMemcpyAsync(&last, offsets + n, sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
      ASL_TRY(sync_stream());
    } else {
      last = offsets[n];
    }
  }
  if (last > 0 && (!mz || !intensity)) return fail(ASL_ERR_INVALID, "encode_entries_batch: null peaks");
  In<float> dmz, din;
  In<int32_t> doff;
  ASL_TRY(dmz.init(mz, (size_t)last));
  ASL_TRY(din.init(intensity, (size_t)last));
  ASL_TRY(doff.init(offsets, (size_t)n + 1));
  ASL_TRY(encode_entries_device(dmz.d, din.d, doff.d, n, min_bound, bin_size, hash_len, seed, norm,
                                reinterpret_cast<uint2 *>(entries), counts, n_over));
  if (dmz.own.p || din.own.p || doff.own.p) ASL_TRY(sync_stream());
  return ASL_OK;
}

}  // extern "C"

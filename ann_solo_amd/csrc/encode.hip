// encode.hip -- feature-hash encoder (replaces spectrum_to_vector,
// /root/reference/src/ann_solo/spectrum.py:146-214).
//
// One 64-lane wavefront per spectrum, four spectra per workgroup. Lanes hash
// their peaks in parallel (float64 NumPy floor-division of the float32 m/z,
// MurmurHash3_x86_32 of the decimal string of the bin index, seed 42); the fp32
// scatter-add into the LDS-resident vector is applied by lane 0 in ascending
// peak order so that colliding bins round exactly like the reference's
// `vector[bin] += intensity` loop. The L2 norm is the canonical ascending-index
// fmaf chain. Output rows are written coalesced.
#include "common.hpp"

namespace asl {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) {
  return (x << r) | (x >> (32 - r));
}

// MurmurHash3_x86_32 over the ASCII decimal representation of v (with '-').
__device__ uint32_t murmur3_decimal(long long v, uint32_t seed) {
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

__global__ __launch_bounds__(64 * ENC_WAVES) void encode_kernel(
    const float *__restrict__ mz, const float *__restrict__ inten,
    const int32_t *__restrict__ offsets, int32_t n, double min_bound, double bin_size,
    int32_t hash_len, uint32_t seed, int norm, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int spec = blockIdx.x * ENC_WAVES + wave;
  // per-wave LDS: vec[hash_len] | hidx[64] | hval[64]
  const int per_wave = hash_len + 128;
  float *vec = reinterpret_cast<float *>(smem) + (size_t)wave * per_wave;
  int *hidx = reinterpret_cast<int *>(vec + hash_len);
  float *hval = vec + hash_len + 64;
  if (spec >= n) return;  // whole wave exits together (no block barriers are used)

  for (int i = lane; i < hash_len; i += 64) vec[i] = 0.0f;
  const int p0 = offsets[spec], p1 = offsets[spec + 1];
  for (int base = p0; base < p1; base += 64) {
    const int p = base + lane;
    if (p < p1) {
      long long b = np_floor_div_bin((double)mz[p] - min_bound, bin_size);
      hidx[lane] = (int)(murmur3_decimal(b, seed) % (uint32_t)hash_len);
      hval[lane] = inten[p];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const int cnt = min(64, p1 - base);
      for (int t = 0; t < cnt; ++t) vec[hidx[t]] += hval[t];
    }
    __builtin_amdgcn_wave_barrier();
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
  float *row = out + (size_t)spec * hash_len;
  for (int i = lane; i < hash_len; i += 64) row[i] = norm ? vec[i] / nrm : vec[i];
}

int encode_device(const float *mz, const float *inten, const int32_t *offsets, int32_t n,
                  double min_bound, double bin_size, int32_t hash_len, uint32_t seed,
                  int norm, float *out) {
  if (n == 0) return ASL_OK;
  ProfScope ps("encode");
  size_t lds = (size_t)ENC_WAVES * (hash_len + 128) * sizeof(float);
  if (lds > 160 * 1024) return fail(ASL_ERR_CAPACITY, "encode: hash_len %d too large for LDS", hash_len);
  dim3 grid((unsigned)cdiv(n, ENC_WAVES));
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)encode_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(encode_kernel, grid, dim3(64 * ENC_WAVES), lds, stream(), mz, inten,
                     offsets, n, min_bound, bin_size, hash_len, seed, norm, out);
  ASL_CHECK_LAUNCH();
  return ASL_OK;
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

}  // extern "C"

// Microbenchmark behind profiles/r04_flat_scan_notes.txt: cost of one accumulator update per lane
// in LDS, random addresses, 24 waves per CU as in the postings scan:
//   0 read - fma - write (the scan's update)   1 ds_add_f32   2 ds_add_u32   3 ds_add_u64
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/lds_atomic_bench.hip -o scripts/tmp/ldsbench.so
// run:   python scripts/lds_atomic_bench.py
#include <hip/hip_runtime.h>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(512, 6) void k(const uint32_t *__restrict__ idx, int iters, float *out) {
  extern __shared__ char smem[];
  float *accf = reinterpret_cast<float *>(smem);
  uint32_t *accu = reinterpret_cast<uint32_t *>(smem);
  unsigned long long *accl = reinterpret_cast<unsigned long long *>(smem);
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 8 * 832 * (MODE == 3 ? 2 : 1); i += 512) accu[i] = 0;
  __syncthreads();
  const int base = wave * 832;
  uint32_t a[8];
  for (int u = 0; u < 8; ++u) a[u] = idx[(blockIdx.x * 512 + tid) * 8 + u] % 832u;
  float q = 1.0f + tid * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t l = base + a[u];
      if (MODE == 0) accf[l] = __builtin_fmaf(q, (float)(a[u] + it), accf[l]);
      if (MODE == 1) __hip_atomic_fetch_add(&accf[l], q * (float)(a[u] + it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 2) __hip_atomic_fetch_add(&accu[l], a[u] * (uint32_t)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 3) __hip_atomic_fetch_add(&accl[l], (unsigned long long)a[u] * (unsigned long long)(it + 12345), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      a[u] = (a[u] * 1664525u + 1013904223u + u) % 832u;
    }
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = accf[base];
}

extern "C" float run(int mode, int iters, int blocks) {
  uint32_t *idx;
  float *out;
  hipMalloc(&idx, (size_t)blocks * 512 * 8 * 4);
  hipMalloc(&out, blocks * 4);
  hipMemset(idx, 0x5a, (size_t)blocks * 512 * 8 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = 8 * 832 * 4 * (mode == 3 ? 2 : 1) + 20000;
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), lds, 0, idx, iters, out);
    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), lds, 0, idx, iters, out);
    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), lds, 0, idx, iters, out);
    if (mode == 3) {
      hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), lds, 0, idx, iters, out);
    }
  };
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(idx);
  hipFree(out);
  return ms;
}

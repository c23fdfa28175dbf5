// DIAGNOSTIC ONLY: does any load cache policy make the L2 fetch less than a whole 128-byte line from HBM?
// Each group of 16 lanes reads the first 64 bytes of a 128-byte line (or of a 256-byte pair); lines are
// visited once, far beyond the caches.  Compare time and FETCH_SIZE across policies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LOADER(NAME, MODS)                                                                       \
  __global__ __launch_bounds__(256) void NAME(const float* __restrict__ src, float* dst, size_t nlines, int stride_b) { \
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;                                    \
    const size_t nthreads = (size_t)gridDim.x * 256;                                              \
    const int sub = threadIdx.x & 15;                                                             \
    float acc = 0.f;                                                                              \
    for (size_t line = gid >> 4; line < nlines; line += nthreads >> 4) {                          \
      const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(src) + line * (size_t)stride_b) + sub; \
      float v;                                                                                    \
      asm volatile("global_load_dword %0, %1, off " MODS "\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); \
      acc += v;                                                                                   \
    }                                                                                             \
    if (acc == 12345.678f) dst[gid] = acc;                                                        \
  }

LOADER(k_plain, "")
LOADER(k_nt, "nt")
LOADER(k_sc0, "sc0")
LOADER(k_sc1, "sc1")
LOADER(k_sc0sc1, "sc0 sc1")
LOADER(k_sc0sc1nt, "sc0 sc1 nt")
LOADER(k_sc1nt, "sc1 nt")

typedef void (*kern_t)(const float*, float*, size_t, int);

int main() {
  const size_t bytes = (size_t)2 << 30;
  float *src, *dst;
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, 1 << 24) != hipSuccess) return 1;
  hipMemset(src, 0, bytes);
  struct { const char* name; kern_t k; } ks[] = {{"plain", k_plain}, {"nt", k_nt}, {"sc0", k_sc0}, {"sc1", k_sc1},
                                                 {"sc0 sc1", k_sc0sc1}, {"sc0 sc1 nt", k_sc0sc1nt}, {"sc1 nt", k_sc1nt}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int stride : {128, 256, 64}) {
    const size_t nlines = bytes / stride;
    for (auto& k : ks) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k.k, dim3(256 * 16), dim3(256), 0, 0, src, dst, nlines, stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("stride %3d B  policy %-11s  %7.3f ms   useful %6.1f GB/s   lines %6.1f GB/s\n", stride, k.name, best,
             nlines * 64.0 / best / 1e6, nlines * (double)(stride < 128 ? 64 : 128) / best / 1e6);
    }
  }
  return 0;
}

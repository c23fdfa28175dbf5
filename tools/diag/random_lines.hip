// DIAGNOSTIC ONLY: HBM read rate for uniformly random, aligned chunks of 128 B .. 2 KB out of a 4 GB buffer
// (every chunk read whole with 16 bytes per lane, 8 loads in flight per lane, 16 waves per CU).
// This is the ceiling a gather kernel can reach at a given contiguity.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

template <int CS>
__global__ __launch_bounds__(256) void k_random(const float4* __restrict__ src, float* dst, size_t nchunks_total, unsigned nchunk_mask, int iters) {
  constexpr int LPC = CS / 16;                 // lanes per chunk
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  const unsigned grp = gid / LPC, sub = gid % LPC;
  float acc = 0.f;
  for (int it = 0; it < iters; it += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const unsigned c = hash32(grp * 977u + (unsigned)(it + u) * 0x9e3779b9u) & nchunk_mask;
      v[u] = src[(size_t)c * LPC + sub];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += v[u].x + v[u].w;
  }
  if (acc == 12345.678f) dst[gid] = acc;
}

template <int CS>
void run(const float4* src, float* dst, size_t bytes) {
  const size_t nchunks = bytes / CS;  // power of two
  const int iters = 64;
  const int grid = 256 * 16;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_random<CS>, dim3(grid), dim3(256), 0, 0, src, dst, nchunks, (unsigned)(nchunks - 1), iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double total = (double)grid * 256 * 16.0 * iters;
  printf("random aligned chunks of %4d B: %7.3f ms  %7.1f GB/s\n", CS, best, total / best / 1e6);
}

int main() {
  const size_t bytes = (size_t)4 << 30;
  float4* src; float* dst;
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, 1 << 24) != hipSuccess) return 1;
  hipMemset(src, 0, bytes);
  run<128>(src, dst, bytes); run<256>(src, dst, bytes); run<512>(src, dst, bytes); run<1024>(src, dst, bytes); run<2048>(src, dst, bytes);
  run<128>(src, dst, bytes); run<256>(src, dst, bytes);
  return 0;
}

// DIAGNOSTIC ONLY: what the memory system delivers for the metric kernel's TRAFFIC MIX, independent of the kernel:
// per 1280 bytes moved, 768 B of uniformly random 128-byte lines out of a 4 GB buffer (the volume gathers: 2598 of
// the 4244 B a pixel-edge moves), 256 B of streaming reads (offsets + coords: 792 B) and 256 B of streaming writes
// (the output: 854 B) — read : write = 4 : 1 as measured by the PMC passes (profiles/traffic_r02_cold_tiled.json).
// Every line is touched once per launch (nothing is served from the Infinity Cache); 16 bytes per lane, 8 lanes per
// line, WAVES waves per CU with 8 loads in flight per lane.  Also prints the read-only rates for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

// MODE 0: mix; 1: random lines only; 2: streaming reads only; 3: random reads + streaming writes (no streaming reads)
template <int MODE>
__global__ __launch_bounds__(256) void k_mix(const float4* __restrict__ rnd, const float4* __restrict__ seq, float4* __restrict__ dst,
                                             unsigned line_mask, size_t seq_line_mask, size_t dst_f4_mask, int iters) {
  // every index is masked with its (power-of-two) buffer size: in bounds by construction; the host sizes the run so
  // that no mask ever wraps
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  const unsigned grp = gid >> 3, sub = gid & 7;
  const size_t ngrp = (size_t)gridDim.x * 32;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; it++) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const bool streaming = MODE == 2 || (MODE == 0 && u >= 6);
      if (MODE == 3 && u >= 6) { v[u] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
      if (streaming) {
        const size_t k = MODE == 2 ? (size_t)it * 8 + u : (size_t)it * 2 + (u - 6);
        const size_t line = (k * ngrp + grp) & seq_line_mask;       // every line once, neighbouring groups neighbouring lines
        v[u] = seq[line * 8 + sub];
      } else {
        const unsigned c = hash32(grp * 977u + (unsigned)(it * 8 + u) * 0x9e3779b9u) & line_mask;
        v[u] = rnd[(size_t)c * 8 + sub];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    if (MODE == 0 || MODE == 3) {   // 256 B per group and iteration, streaming
      const size_t o = (((size_t)it * ngrp + grp) * 16 + sub * 2) & dst_f4_mask;
      dst[o] = acc;
      dst[o + 1] = acc;
    }
  }
  if (MODE != 0 && MODE != 3 && acc.x == 12345.678f) dst[gid & dst_f4_mask] = acc;
}

template <int MODE>
void run(const char* name, const float4* rnd, const float4* seq, float4* dst, size_t rnd_bytes, size_t seq_bytes, size_t dst_bytes,
         int waves_per_cu, int iters) {
  const int grid = 256 * waves_per_cu / 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix<MODE>, dim3(grid), dim3(256), 0, 0, rnd, seq, dst, (unsigned)(rnd_bytes / 128 - 1), seq_bytes / 128 - 1,
                       dst_bytes / 16 - 1, iters);
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double groups = (double)grid * 32;
  const double rd = groups * iters * (MODE == 0 ? 1024.0 : MODE == 3 ? 768.0 : 1024.0), wr = (MODE == 0 || MODE == 3) ? groups * iters * 256.0 : 0.0;
  printf("%-44s waves/CU %2d: %7.3f ms  read %6.1f + write %6.1f = %6.1f GB/s\n", name, waves_per_cu, best, rd / best / 1e6, wr / best / 1e6,
         (rd + wr) / best / 1e6);
}

int main() {
  const size_t rnd_bytes = (size_t)4 << 30, seq_bytes = (size_t)2 << 30, dst_bytes = (size_t)2 << 30;
  float4 *rnd, *seq, *dst;
  if (hipMalloc(&rnd, rnd_bytes) != hipSuccess || hipMalloc(&seq, seq_bytes) != hipSuccess || hipMalloc(&dst, dst_bytes) != hipSuccess) return 1;
  hipMemset(rnd, 0, rnd_bytes); hipMemset(seq, 0, seq_bytes); hipMemset(dst, 0, dst_bytes);
  for (int w : {8, 16, 32}) {
    // iterations sized so that the streaming buffers are walked at most once (streaming-only mode: 1 KB per group and
    // iteration out of 2 GB; the mix: 256 B read + 256 B written per group and iteration)
    const size_t ngrp = (size_t)(256 * w / 4) * 32;
    const int it_mix = (int)(dst_bytes / (ngrp * 256) > 96 ? 96 : dst_bytes / (ngrp * 256));
    const int it_seq = (int)(seq_bytes / (ngrp * 1024) > 96 ? 96 : seq_bytes / (ngrp * 1024));
    run<1>("random 128-byte lines, reads only", rnd, seq, dst, rnd_bytes, seq_bytes, dst_bytes, w, it_mix);
    run<2>("streaming reads only", rnd, seq, dst, rnd_bytes, seq_bytes, dst_bytes, w, it_seq);
    run<3>("random lines + streaming writes (3:1)", rnd, seq, dst, rnd_bytes, seq_bytes, dst_bytes, w, it_mix);
    run<0>("metric-kernel mix (random 3 : stream 1 : write 1)", rnd, seq, dst, rnd_bytes, seq_bytes, dst_bytes, w, it_mix);
  }
  return 0;
}

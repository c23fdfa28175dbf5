// lgu_common.hpp — shared device/host helpers for the gfx950 correlation-sampling kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/lgu_corr.h"

namespace lgu {

constexpr int kWave = 64;  // CDNA wavefront width; every kernel here is written for it

__device__ __forceinline__ bool in_bounds(int h, int w, int H, int W) {
  return h >= 0 && h < H && w >= 0 && w < W;
}

// Bilinear blend in the reference's evaluation order (defCorrSample_kernel.cu:83-86):
// the four weights are formed in fp32 first, then Q11*w11 + Q21*w21 + Q12*w12 + Q22*w22
// summed left to right. Built with -ffp-contract=off so no FMA contraction changes it.
__device__ __forceinline__ float bilerp(float q11, float q21, float q12, float q22, float dx, float dy) {
  const float w11 = (1.0f - dy) * (1.0f - dx);
  const float w21 = (1.0f - dy) * dx;
  const float w12 = dy * (1.0f - dx);
  const float w22 = dy * dx;
  return q11 * w11 + q21 * w21 + q12 * w12 + q22 * w22;
}

// ---- wave64 all-reduce (min / max / sum) ------------------------------------------
// Butterfly over the 64 lanes; every lane ends with the result.
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(v, m, kWave);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const int o = __shfl_xor(v, m, kWave);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? LGU_OK : (int)e;
}

inline int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

}  // namespace lgu

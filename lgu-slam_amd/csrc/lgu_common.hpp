// lgu_common.hpp — shared device/host helpers for the gfx950 correlation-sampling kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>

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

// ---- packed int16 pairs (x low, y high): one v_pk_min_i16 / v_pk_max_i16 serves both axes
typedef short __attribute__((ext_vector_type(2))) short2v;

__device__ __forceinline__ int pk16(int x, int y) { return (x & 0xffff) | (y << 16); }
__device__ __forceinline__ int pk_lo(int v) { return (int)(short)(v & 0xffff); }
__device__ __forceinline__ int pk_hi(int v) { return v >> 16; }
__device__ __forceinline__ int pk_min(int a, int b) {
  return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ int pk_max(int a, int b) {
  return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Wave-wide packed min/max: DPP inside each row of 16 lanes (xor 1, xor 2, half-mirror,
// mirror), then the four row results are read with v_readlane and combined (wave-uniform).
template <bool IS_MIN>
__device__ __forceinline__ int wave_pk_reduce(int v) {
#define LGU_DPP_STEP(ctrl)                                                  \
  {                                                                         \
    const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xf, 0xf, false); \
    v = IS_MIN ? pk_min(v, o) : pk_max(v, o);                               \
  }
  LGU_DPP_STEP(0xB1)   // quad_perm:[1,0,3,2]
  LGU_DPP_STEP(0x4E)   // quad_perm:[2,3,0,1]
  LGU_DPP_STEP(0x141)  // row_half_mirror
  LGU_DPP_STEP(0x140)  // row_mirror
#undef LGU_DPP_STEP
  const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  return IS_MIN ? pk_min(pk_min(r0, r1), pk_min(r2, r3)) : pk_max(pk_max(r0, r1), pk_max(r2, r3));
}

// Sum over each row of 16 lanes (every lane of the row gets the row total).
__device__ __forceinline__ float row16_sum(float v) {
#define LGU_SUM_STEP(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false));
  LGU_SUM_STEP(0xB1)
  LGU_SUM_STEP(0x4E)
  LGU_SUM_STEP(0x141)
  LGU_SUM_STEP(0x140)
#undef LGU_SUM_STEP
  return v;
}

inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? LGU_OK : (int)e;
}

// ---- debug knobs -------------------------------------------------------------------
// The LGU_* environment variables read through env_int() select superseded kernels and A/B settings for the tests and
// tools/.  They are honoured ONLY when LGU_DEBUG_KNOBS=1 was in the environment when the library was loaded (read once,
// capi.hip; lgu_debug_knobs_enabled() reports it).  Otherwise env_int() returns the default without touching the
// environment: a production launch makes no getenv call and a stray LGU_* variable changes nothing.
bool debug_knobs();   // capi.hip

inline int env_int(const char* name, int dflt) {
  if (!debug_knobs()) return dflt;
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

// ---- per-device launch state ---------------------------------------------------------
// A process may drive several devices.  What the launchers cache — "this kernel may use more than 64 KB of dynamic LDS"
// (an attribute of the function ON A DEVICE) and the device's CU count — is therefore kept per device, in namespace-scope
// tables (no function-local statics); racing writers store the same values.
constexpr int kMaxDevices = 64;

inline int current_device() {
  int d = -1;
  return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < kMaxDevices) ? d : -1;
}

struct DeviceFlags { std::atomic<unsigned char> done[kMaxDevices]; };
template <auto Kern> inline DeviceFlags g_lds_attr{};   // one table per kernel instantiation

template <auto Kern>
inline hipError_t allow_max_dynamic_lds() {
  const int d = current_device();
  if (d >= 0 && g_lds_attr<Kern>.done[d].load(std::memory_order_relaxed)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess && d >= 0) g_lds_attr<Kern>.done[d].store(1, std::memory_order_relaxed);
  return e;
}

inline std::atomic<int> g_cu_count[kMaxDevices];

inline int device_cu_count() {   // CUs of the current device (256 on an MI355X; also the answer when the query fails)
  const int d = current_device();
  if (d >= 0) {
    const int c = g_cu_count[d].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int cus = 0;
  if (d < 0 || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cus < 1) return 256;
  g_cu_count[d].store(cus, std::memory_order_relaxed);
  return cus;
}

// LGU_LDS_PAD (occupancy experiments only, tools/ab_cold.py): extra dynamic LDS per workgroup, clamped to [0, 96 KiB]
inline size_t lds_pad() {
  const int v = env_int("LGU_LDS_PAD", 0);
  return (size_t)(v < 0 ? 0 : (v > 96 * 1024 ? 96 * 1024 : v));
}

}  // namespace lgu

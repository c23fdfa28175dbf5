// defcorr_bwd.hip — backward of the deformable / plain volume sampler (training only).
//
// Replaces (reference, relative to /root/reference):
//   offersample_LGS/defCorrSample_kernel.cu:93-162  defCorr_index_backward_kernel
//   offersample_LGS/corrSample_kernel.cu:84-136     corr_index_backward_kernel (LGU variant)
//
// One WAVE per (edge, pixel): the pixel's slice of volume_grad is private to that wave, so
// the scatter needs no global atomics.  Lane t owns tap t; the four bilinear contributions
// of every tap are accumulated in an LDS image of the tap box with LDS float atomics
// (taps of one pixel can hit the same element), then the box is added to the
// caller-zeroed volume_grad with row-contiguous read-modify-writes.  Taps whose box does
// not fit LDS take global atomics (never the case for |offset| < 4).
// The accumulation order inside one element differs from the reference's sequential
// i,j loop; the result agrees to fp32 rounding (tests: 1e-5).
#include "lgu_common.hpp"

namespace lgu {

constexpr int BW_WAVES = 4;        // waves (= pixels) per workgroup
constexpr int BW_BOX_FLOATS = 1024;  // per-wave LDS image: up to 1024 floats (e.g. 32 x 32)

template <bool HAS_OFFSET>
__global__ __launch_bounds__(BW_WAVES * kWave) void defcorr_bwd_kernel(
    const float* __restrict__ volume, const float* __restrict__ coords, float* offset,
    const float* __restrict__ corr_grad, float* volume_grad, float* __restrict__ offset_grad, int E, int H1, int W1,
    int H2, int W2, int r) {
  __shared__ float box_all[BW_WAVES][BW_BOX_FLOATS];
  const int lane = threadIdx.x & (kWave - 1);
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* const box = box_all[w];
  const int rd = 2 * r + 1, nt = rd * rd;
  const size_t HW1 = (size_t)H1 * W1, HW2 = (size_t)H2 * W2;
  const size_t npix = (size_t)E * HW1;
  const size_t pix = (size_t)blockIdx.x * BW_WAVES + w;
  if (pix >= npix) return;  // wave-uniform; no block-level barrier below
  const int e = (int)(pix / HW1);
  const size_t yx = pix - (size_t)e * HW1;
  const float x0 = coords[((size_t)e * 2 + 0) * HW1 + yx];
  const float y0 = coords[((size_t)e * 2 + 1) * HW1 + yx];
  const float* V = volume + pix * HW2;
  float* VG = volume_grad + pix * HW2;

  for (int t0 = 0; t0 < nt; t0 += kWave) {  // rd*rd may exceed 64 for radius >= 4
    const int t = t0 + lane;
    const bool tap = t < nt;
    const int i = tap ? t / rd : 0, j = tap ? t - i * rd : 0;
    float ox = 0.0f, oy = 0.0f;
    if (HAS_OFFSET && tap) {
      float* op = offset + (pix * nt + t) * 2;
      if (i == r && j == r) {
        op[0] = 0.0f;  // defCorrSample_kernel.cu:122-123
        op[1] = 0.0f;
      } else {
        ox = op[0];
        oy = op[1];
      }
    }
    const float ofsX = ox + x0, ofsY = oy + y0;
    const int fx = (int)floorf(ofsX), fy = (int)floorf(ofsY);
    const float dx = ofsX - (float)fx, dy = ofsY - (float)fy;
    const int x1 = fx - r + i, y1 = fy - r + j;
    const bool valid = tap && in_bounds(y1, x1, H2, W2);
    const bool xin = x1 + 1 < W2, yin = y1 + 1 < H2;
    const int xh = xin ? x1 + 1 : x1, yh = yin ? y1 + 1 : y1;
    const int xlo = wave_min_i32(valid ? x1 : 0x7fffffff), ylo = wave_min_i32(valid ? y1 : 0x7fffffff);
    const int xhi = wave_max_i32(valid ? xh : (int)0x80000000), yhi = wave_max_i32(valid ? yh : (int)0x80000000);
    float g = 0.0f, q11 = 0.0f, q21 = 0.0f, q12 = 0.0f, q22 = 0.0f;
    if (valid) {
      g = corr_grad[((size_t)e * nt + t) * HW1 + yx];
      if (HAS_OFFSET) {  // the volume only feeds offset_grad
        const float* s = V + (size_t)y1 * W2 + x1;
        q11 = s[0];
        if (xin) q21 = s[1];
        if (yin) q12 = s[W2];
        if (xin && yin) q22 = s[W2 + 1];
      }
    }
    const float w11 = ((1.0f - dy) * (1.0f - dx)) * g, w21 = ((1.0f - dy) * dx) * g;
    const float w12 = (dy * (1.0f - dx)) * g, w22 = (dy * dx) * g;
    if (HAS_OFFSET && tap) {
      float* og = offset_grad + (pix * nt + t) * 2;
      float gy = 0.0f, gx = 0.0f;
      if (valid) {  // :156-157
        gy = (-q11 * (1.0f - dx) - q21 * dx + q12 * (1.0f - dx) + q22 * dx) * g;
        gx = (-q11 * (1.0f - dy) + q21 * (1.0f - dy) - q12 * dy + q22 * dy) * g;
      }
      og[0] = gx;
      og[1] = gy;
    }
    if (yhi < ylo) continue;
    const int bw = xhi - xlo + 1, bh = yhi - ylo + 1;
    if (bw * bh <= BW_BOX_FLOATS) {
      for (int k = lane; k < bw * bh; k += kWave) box[k] = 0.0f;
      __builtin_amdgcn_wave_barrier();
      if (valid) {
        float* b = box + (y1 - ylo) * bw + (x1 - xlo);
        atomicAdd(b, w11);
        if (xin) atomicAdd(b + 1, w21);
        if (yin) atomicAdd(b + bw, w12);
        if (xin && yin) atomicAdd(b + bw + 1, w22);
      }
      __builtin_amdgcn_wave_barrier();
      for (int k = lane; k < bw * bh; k += kWave) {
        const int by = k / bw, bx = k - by * bw;
        const float v = box[k];
        if (v != 0.0f) VG[(size_t)(ylo + by) * W2 + (xlo + bx)] += v;
      }
      __builtin_amdgcn_wave_barrier();
    } else if (valid) {
      float* d = VG + (size_t)y1 * W2 + x1;
      atomicAdd(d, w11);
      if (xin) atomicAdd(d + 1, w21);
      if (yin) atomicAdd(d + W2, w12);
      if (xin && yin) atomicAdd(d + W2 + 1, w22);
    }
    __threadfence_block();  // next pass (rd*rd > 64 only) may touch the same elements from other lanes
  }
}

static int bwd_launch(const float* volume, const float* coords, float* offset, const float* corr_grad,
                      float* volume_grad, float* offset_grad, int E, int H1, int W1, int H2, int W2, int radius,
                      void* stream, bool has_offset) {
  if (!coords || !corr_grad || !volume_grad) return LGU_E_BADARG;
  if (has_offset && (!volume || !offset || !offset_grad)) return LGU_E_BADARG;
  if (E < 0 || H1 < 1 || W1 < 1 || H2 < 1 || W2 < 1 || radius < 0 || radius > LGU_MAX_RADIUS) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  const size_t npix = (size_t)E * H1 * W1;
  const unsigned grid = (unsigned)((npix + BW_WAVES - 1) / BW_WAVES);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (has_offset)
    hipLaunchKernelGGL(defcorr_bwd_kernel<true>, dim3(grid), dim3(BW_WAVES * kWave), 0, st, volume, coords, offset,
                       corr_grad, volume_grad, offset_grad, E, H1, W1, H2, W2, radius);
  else  // the plain sampler's backward never reads the volume (corrSample_kernel.cu:84-136)
    hipLaunchKernelGGL(defcorr_bwd_kernel<false>, dim3(grid), dim3(BW_WAVES * kWave), 0, st, (const float*)nullptr, coords,
                       (float*)nullptr, corr_grad, volume_grad, (float*)nullptr, E, H1, W1, H2, W2, radius);
  return launch_status();
}

}  // namespace lgu

extern "C" {

int lgu_defcorr_bwd_f32(const float* volume, const float* coords, float* offset, const float* corr_grad,
                        float* volume_grad, float* offset_grad, int E, int H1, int W1, int H2, int W2, int radius,
                        void* stream) {
  return lgu::bwd_launch(volume, coords, offset, corr_grad, volume_grad, offset_grad, E, H1, W1, H2, W2, radius,
                         stream, true);
}

int lgu_corridx_bwd_f32(const float* volume, const float* coords, const float* corr_grad, float* volume_grad, int E,
                        int H1, int W1, int H2, int W2, int radius, void* stream) {
  (void)volume;
  return lgu::bwd_launch(nullptr, coords, nullptr, corr_grad, volume_grad, nullptr, E, H1, W1, H2, W2, radius, stream,
                         false);
}

}  // extern "C"

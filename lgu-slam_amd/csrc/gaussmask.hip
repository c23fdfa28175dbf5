// gaussmask.hip — learnable per-pixel 2-D Gaussian window on the level-0 correlation volume.
//
// Replaces (reference, relative to /root/reference):
//   offersample_LGS/gaussianAttn.cu:19-68    gaussianMask_kernel           (host :134-163)
//   offersample_LGS/gaussianAttn.cu:72-131   gaussianMask_kernel_backward  (host :165-200)
//
// Forward is write-bound: the op's contract is a full zero-filled copy-shaped output with
// a (2R+1)^2 window of re-weighted values per pixel slice.  The reference memsets the
// output (torch::zeros_like) and then scatters 81 4-byte stores per thread with a 12 KiB
// lane stride.  Here ONE pass writes every slice exactly once with 16-byte coalesced
// stores: a workgroup owns SLICES consecutive slices, each thread produces whole float4
// granules and only the granules that intersect the window read the input volume.
#include "lgu_common.hpp"

namespace lgu {

// exp(f1) of gaussianAttn.cu:59-62 in the reference's evaluation order.  The reference's
// `-0.5*(...)` promotes the float sum to double, scales by 0.5 (exact) and narrows back:
// identical to a float multiply, so no fp64 is needed here.
__device__ __forceinline__ float gauss_e(int x1, int y1, float mx, float my, float c1, float c2) {
  const float ddx = (float)x1 - mx, ddy = (float)y1 - my;
  const float temp1 = ddx / c1, temp2 = ddy / c2;
  const float f1 = -0.5f * (temp1 * ddx + temp2 * ddy);
  return expf(f1);
}

constexpr int GM_THREADS = 256;

__global__ __launch_bounds__(GM_THREADS) void gaussmask_fwd_kernel(const float* __restrict__ means,
                                                                   const float* __restrict__ covs,
                                                                   const float* __restrict__ volume,
                                                                   float* __restrict__ volume1, size_t npix, int H2,
                                                                   int W2, int r, int slices_per_block) {
  const int HW2 = H2 * W2;
  const int g_per_slice = HW2 >> 2;  // W2 % 4 == 0 on this path
  const int g_per_row = W2 >> 2;
  const size_t pix0 = (size_t)blockIdx.x * slices_per_block;
  for (int s = 0; s < slices_per_block; s++) {
    const size_t pix = pix0 + s;
    if (pix >= npix) return;
    const float mx = means[pix * 2 + 0], my = means[pix * 2 + 1];  // wave-uniform
    const float c1 = covs[pix * 2 + 0], c2 = covs[pix * 2 + 1];
    const int cx = (int)floorf(mx), cy = (int)floorf(my);
    const int xa = cx - r, xb = cx + r, ya = cy - r, yb = cy + r;  // window, inclusive
    const float4* vin = reinterpret_cast<const float4*>(volume + pix * (size_t)HW2);
    float4* vout = reinterpret_cast<float4*>(volume1 + pix * (size_t)HW2);
    for (int gi = threadIdx.x; gi < g_per_slice; gi += GM_THREADS) {
      const int row = gi / g_per_row;
      const int x4 = (gi - row * g_per_row) << 2;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row >= ya && row <= yb && x4 + 3 >= xa && x4 <= xb) {
        const float4 v = vin[gi];
        // :65  volume * 3 * exp_comp, left to right
        if (x4 + 0 >= xa && x4 + 0 <= xb) o.x = v.x * 3.0f * gauss_e(x4 + 0, row, mx, my, c1, c2);
        if (x4 + 1 >= xa && x4 + 1 <= xb) o.y = v.y * 3.0f * gauss_e(x4 + 1, row, mx, my, c1, c2);
        if (x4 + 2 >= xa && x4 + 2 <= xb) o.z = v.z * 3.0f * gauss_e(x4 + 2, row, mx, my, c1, c2);
        if (x4 + 3 >= xa && x4 + 3 <= xb) o.w = v.w * 3.0f * gauss_e(x4 + 3, row, mx, my, c1, c2);
      }
      vout[gi] = o;
    }
  }
}

// Any W2 / unaligned buffers: one thread per output element.
__global__ __launch_bounds__(256) void gaussmask_fwd_generic_kernel(const float* __restrict__ means,
                                                                    const float* __restrict__ covs,
                                                                    const float* __restrict__ volume,
                                                                    float* __restrict__ volume1, size_t npix, int H2,
                                                                    int W2, int r) {
  const size_t HW2 = (size_t)H2 * W2;
  const size_t total = npix * HW2;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = idx / HW2;
    const int rem = (int)(idx - pix * HW2);
    const int y1 = rem / W2, x1 = rem - y1 * W2;
    const float mx = means[pix * 2 + 0], my = means[pix * 2 + 1];
    const int cx = (int)floorf(mx), cy = (int)floorf(my);
    float o = 0.0f;
    if (x1 >= cx - r && x1 <= cx + r && y1 >= cy - r && y1 <= cy + r)
      o = volume[idx] * 3.0f * gauss_e(x1, y1, mx, my, covs[pix * 2 + 0], covs[pix * 2 + 1]);
    volume1[idx] = o;
  }
}

// Backward: one wave per pixel, lanes over the window taps, sums reduced across the wave.
// The reference accumulates the (2R+1)^2 terms sequentially per thread; a tree sum differs
// from that only by fp32 rounding (tests: 1e-5 relative to the gradient scale).
// The two `0.5` literals of gaussianAttn.cu:119,121 make those products double
// expressions in the reference; they are evaluated in double here as well.
__global__ __launch_bounds__(256) void gaussmask_bwd_kernel(const float* __restrict__ means,
                                                            const float* __restrict__ covs,
                                                            const float* __restrict__ volume,
                                                            const float* __restrict__ volume1_grad,
                                                            float* __restrict__ means_grad,
                                                            float* __restrict__ covs_grad, size_t npix, int H2, int W2,
                                                            int r) {
  const int lane = threadIdx.x & (kWave - 1);
  const size_t pix = (size_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const int rd = 2 * r + 1, nt = rd * rd;
  const float mx = means[pix * 2 + 0], my = means[pix * 2 + 1];
  const float c1 = covs[pix * 2 + 0], c2 = covs[pix * 2 + 1];
  const int cx = (int)floorf(mx), cy = (int)floorf(my);
  const float* V = volume + pix * (size_t)H2 * W2;
  const float* G = volume1_grad + pix * (size_t)H2 * W2;
  float mg0 = 0.f, mg1 = 0.f, cg0 = 0.f, cg1 = 0.f;
  for (int t = lane; t < nt; t += kWave) {
    const int i = t / rd, j = t - i * rd;
    const int x1 = cx - r + i, y1 = cy - r + j;
    if (in_bounds(y1, x1, H2, W2)) {
      const float ddx = (float)x1 - mx, ddy = (float)y1 - my;
      const float e = gauss_e(x1, y1, mx, my, c1, c2);
      const float v = V[(size_t)y1 * W2 + x1], g = G[(size_t)y1 * W2 + x1];
      mg0 += 3.0f * v * (e * ddx / c1) * g;  // :116
      mg1 += 3.0f * v * (e * ddy / c2) * g;  // :117
      const float dE1 = (float)((double)e * 0.5 * (double)ddx * (double)ddx / (double)(c1 * c1));  // :119
      const float dE2 = (float)((double)e * 0.5 * (double)ddy * (double)ddy / (double)(c2 * c2));  // :121
      cg0 += (3.0f * v * dE1) * g;  // :124
      cg1 += (3.0f * v * dE2) * g;  // :125
    }
  }
  mg0 = wave_sum_f32(mg0);
  mg1 = wave_sum_f32(mg1);
  cg0 = wave_sum_f32(cg0);
  cg1 = wave_sum_f32(cg1);
  if (lane == 0) {
    means_grad[pix * 2 + 0] = mg0;
    means_grad[pix * 2 + 1] = mg1;
    covs_grad[pix * 2 + 0] = cg0;
    covs_grad[pix * 2 + 1] = cg1;
  }
}


// ---- fused volume post-processing ("next" row f1 of SURVEY.md §8) -----------------------
// One pass over every level-0 slice replaces, per new edge, the reference's
//   gaussianMask kernel + zeros_like memset            (gaussianAttn.cu:134-163)
//   corr1 / (6.28*sqrt(det)) + corr                    (gaussianMask_cuda.py:85-86, 2 torch passes)
//   3x avg_pool2d(2, stride 2) over the target dims    (corr.py:83-86)
// i.e. ~100 KB of HBM traffic per pixel, with 12 KB read + 16 KB written: the slice is held in
// LDS, re-weighted in place, written out as level 0 and pooled to levels 1..L-1 from LDS.
// Arithmetic order follows the torch ops it replaces: t = (v*3*e)/den + v, pooled value =
// (((a00 + a01) + a10) + a11) / 4 (ATen's avg_pool2d accumulation order).
constexpr int VP_THREADS = 256;
constexpr int VP_MAXL = 4;

struct VolPyrParams {
  const float* means;
  const float* covs;
  const void* vin;      // fp32, or half when the kernel's HALF_IN is set
  const void* det;      // optional (npix): det the caller's Gaussian head produced (fp32, or half when det_half); null = cov0 * cov1
  int det_half;
  float* out[VP_MAXL];  // out[0] may alias an fp32 vin
  size_t npix;
  int H2, W2, L, r;
};

// TILED: every level is written in the tiled slice layout of include/lgu_corr.h (LGU_PYR_TILED): 4 x 8 element
// tiles, one 128-byte line each.  The slice is complete in LDS before anything is written, so level 0 may
// still be converted in place when its padded size equals H2*W2.
__device__ __forceinline__ int tiled_pos(int y, int x, int tpr) {
  return (((y >> 2) * tpr + (x >> 3)) << 5) + ((y & 3) << 3) + (x & 7);
}

// HALF_IN: the raw volume is IEEE half — what the all-pairs matmul of half feature maps returns (corr.py:145-152 on
// depth_video's half fmaps); the `.float()` of corr.py:64 is then this kernel's load (exact), not a 1.1 GB pass of its own.
template <bool TILED, bool HALF_IN>
__global__ __launch_bounds__(VP_THREADS) void volume_pyramid_kernel(const VolPyrParams p) {
  extern __shared__ float4 vp_smem4[];
  float* const sm = reinterpret_cast<float*>(vp_smem4);
  const size_t pix = blockIdx.x;
  const int H2 = p.H2, W2 = p.W2, HW2 = H2 * W2;
  const float mx = p.means[pix * 2 + 0], my = p.means[pix * 2 + 1];
  const float c1 = p.covs[pix * 2 + 0], c2 = p.covs[pix * 2 + 1];
  // denominator = 6.28 * torch.sqrt(det) (gaussianMask_cuda.py:79,85; det = cov0*cov1).  Under autocast the reference's
  // det is a HALF tensor (factor_graph.py:90 builds CorrBlock inside autocast): sqrt and the product with the Python
  // scalar are then half kernels — evaluated in fp32, rounded to half after each — and the fp32 division promotes the
  // rounded value back.  A half `det` reproduces exactly that; an fp32 `det` (or none) is the fp32 evaluation.
  float den;
  if (p.det && p.det_half) {
    const _Float16 sq = (_Float16)sqrtf((float)static_cast<const _Float16*>(p.det)[pix]);
    // torch's half multiply rounds the fp32 product to fp32 and THEN to half.  Left to itself the compiler fuses the
    // product and the conversion into v_fma_mixlo_f16, which rounds the exact product once — different on ties
    // (det = 4.203125: 2.05078125 * 6.28f is within 4e-7 of the midpoint of two halves).  The empty asm pins the
    // fp32 product in a register.
    float prod = (float)sq * 6.28f;
    asm volatile("" : "+v"(prod));
    den = (float)(_Float16)prod;
  } else {
    den = 6.28f * sqrtf(p.det ? static_cast<const float*>(p.det)[pix] : c1 * c2);
  }
  const int cx = (int)floorf(mx), cy = (int)floorf(my);
  const int xa = cx - p.r, xb = cx + p.r, ya = cy - p.r, yb = cy + p.r;
  typedef _Float16 vp_half4 __attribute__((ext_vector_type(4)));
  const float4* vin = reinterpret_cast<const float4*>(static_cast<const float*>(p.vin) + (HALF_IN ? 0 : pix * (size_t)HW2));
  const vp_half4* vinh = reinterpret_cast<const vp_half4*>(static_cast<const _Float16*>(p.vin) + (HALF_IN ? pix * (size_t)HW2 : 0));
  const int tpr0 = (W2 + 7) >> 3;
  const int ssz0 = TILED ? ((H2 + 3) >> 2) * tpr0 * 32 : HW2;
  float4* vout = reinterpret_cast<float4*>(p.out[0] + pix * (size_t)ssz0);
  const int g_per_row = W2 >> 2;
  for (int gi = threadIdx.x; gi < (HW2 >> 2); gi += VP_THREADS) {
    const int row = gi / g_per_row;
    const int x4 = (gi - row * g_per_row) << 2;
    float4 v;
    if constexpr (HALF_IN) {
      const vp_half4 hv = vinh[gi];
      v = make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
    } else {
      v = vin[gi];
    }
    if (row >= ya && row <= yb && x4 + 3 >= xa && x4 <= xb) {
      // inside the window: corr1 / denominator + corr; outside corr1 == 0 and the sum is v exactly
      if (x4 + 0 >= xa && x4 + 0 <= xb) v.x = (v.x * 3.0f * gauss_e(x4 + 0, row, mx, my, c1, c2)) / den + v.x;
      if (x4 + 1 >= xa && x4 + 1 <= xb) v.y = (v.y * 3.0f * gauss_e(x4 + 1, row, mx, my, c1, c2)) / den + v.y;
      if (x4 + 2 >= xa && x4 + 2 <= xb) v.z = (v.z * 3.0f * gauss_e(x4 + 2, row, mx, my, c1, c2)) / den + v.z;
      if (x4 + 3 >= xa && x4 + 3 <= xb) v.w = (v.w * 3.0f * gauss_e(x4 + 3, row, mx, my, c1, c2)) / den + v.w;
    }
    reinterpret_cast<float4*>(sm)[gi] = v;
    if (!TILED) vout[gi] = v;
  }
  if (TILED) {
    __syncthreads();  // every read of vin is done: out[0] may alias it
    for (int gi = threadIdx.x; gi < (ssz0 >> 2); gi += VP_THREADS) {
      const int t = gi << 2, tile = t >> 5, ty = tile / tpr0, tx = tile - ty * tpr0;
      const int y = ty * 4 + ((t & 31) >> 3), x = tx * 8 + (t & 7);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y < H2 && x < W2) v = *reinterpret_cast<const float4*>(sm + y * W2 + x);  // W2 % 4 == 0
      vout[gi] = v;
    }
  }
  float* src = sm;
  int Hs = H2, Ws = W2;
  for (int l = 1; l < p.L; l++) {
    __syncthreads();
    const int Hd = Hs >> 1, Wd = Ws >> 1;
    float* dst = src + Hs * Ws;
    if (!TILED) {
      float* gout = p.out[l] + pix * (size_t)(Hd * Wd);
      for (int idx = threadIdx.x; idx < Hd * Wd; idx += VP_THREADS) {
        const int y = idx / Wd, x = idx - y * Wd;
        const float* s = src + (2 * y) * Ws + 2 * x;
        const float o = (((s[0] + s[1]) + s[Ws]) + s[Ws + 1]) / 4.0f;
        dst[idx] = o;
        gout[idx] = o;
      }
    } else {
      const int tpr = (Wd + 7) >> 3, ssz = ((Hd + 3) >> 2) * tpr * 32;
      float* gout = p.out[l] + pix * (size_t)ssz;
      for (int t = threadIdx.x; t < ssz; t += VP_THREADS) {  // tiled order: coalesced stores
        const int tile = t >> 5, ty = tile / tpr, tx = tile - ty * tpr;
        const int y = ty * 4 + ((t & 31) >> 3), x = tx * 8 + (t & 7);
        float o = 0.0f;
        if (y < Hd && x < Wd) {
          const float* s = src + (2 * y) * Ws + 2 * x;
          o = (((s[0] + s[1]) + s[Ws]) + s[Ws + 1]) / 4.0f;
          dst[y * Wd + x] = o;
        }
        gout[t] = o;
      }
    }
    src = dst;
    Hs = Hd;
    Ws = Wd;
  }
}

// Layout conversion of whole slices (setup / tests; not on the lookup path).
__global__ __launch_bounds__(256) void volume_retile_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            size_t nslices, int H2, int W2, int to_tiled) {
  const int tpr = (W2 + 7) >> 3;
  const size_t ssz = (size_t)((H2 + 3) >> 2) * tpr * 32, rsz = (size_t)H2 * W2;
  const size_t total = nslices * (to_tiled ? ssz : rsz);
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    if (to_tiled) {
      const size_t sl = idx / ssz;
      const int t = (int)(idx - sl * ssz), tile = t >> 5, ty = tile / tpr, tx = tile - ty * tpr;
      const int y = ty * 4 + ((t & 31) >> 3), x = tx * 8 + (t & 7);
      dst[idx] = (y < H2 && x < W2) ? src[sl * rsz + (size_t)y * W2 + x] : 0.0f;
    } else {
      const size_t sl = idx / rsz;
      const int r = (int)(idx - sl * rsz), y = r / W2, x = r - y * W2;
      dst[idx] = src[sl * ssz + tiled_pos(y, x, tpr)];
    }
  }
}

}  // namespace lgu

extern "C" {

int lgu_gaussmask_fwd_f32(const float* means, const float* covs, const float* volume, float* volume1, int E, int H1,
                          int W1, int H2, int W2, int radius, void* stream) {
  using namespace lgu;
  if (!means || !covs || !volume || !volume1) return LGU_E_BADARG;
  if (E < 0 || H1 < 1 || W1 < 1 || H2 < 1 || W2 < 1 || radius < 0) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t npix = (size_t)E * H1 * W1;
  const bool fast = (W2 % 4 == 0) && ((reinterpret_cast<uintptr_t>(volume) | reinterpret_cast<uintptr_t>(volume1)) & 15) == 0;
  if (fast) {
    // a 256-thread block writes >= 16 KiB: enough stores in flight per block, few blocks idle
    const int g_per_slice = (H2 * W2) >> 2;
    int spb = (4 * GM_THREADS + g_per_slice - 1) / g_per_slice;
    if (spb < 1) spb = 1;
    const size_t grid = (npix + spb - 1) / spb;
    hipLaunchKernelGGL(gaussmask_fwd_kernel, dim3((unsigned)grid), dim3(GM_THREADS), 0, st, means, covs, volume,
                       volume1, npix, H2, W2, radius, spb);
  } else {
    const size_t total = npix * H2 * W2;
    const size_t want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < 65536u * 8 ? want : 65536u * 8);
    hipLaunchKernelGGL(gaussmask_fwd_generic_kernel, dim3(grid), dim3(256), 0, st, means, covs, volume, volume1, npix,
                       H2, W2, radius);
  }
  return launch_status();
}

int lgu_gaussmask_bwd_f32(const float* means, const float* covs, const float* volume, const float* volume1_grad,
                          float* means_grad, float* covs_grad, int E, int H1, int W1, int H2, int W2, int radius,
                          void* stream) {
  using namespace lgu;
  if (!means || !covs || !volume || !volume1_grad || !means_grad || !covs_grad) return LGU_E_BADARG;
  if (E < 0 || H1 < 1 || W1 < 1 || H2 < 1 || W2 < 1 || radius < 0) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  const size_t npix = (size_t)E * H1 * W1;
  const unsigned grid = (unsigned)((npix + 3) / 4);
  hipLaunchKernelGGL(gaussmask_bwd_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), means,
                     covs, volume, volume1_grad, means_grad, covs_grad, npix, H2, W2, radius);
  return launch_status();
}

static int volume_pyramid_host(const float* means, const float* covs, const void* volume, float* const* levels, int L,
                               int E, int H1, int W1, int H2, int W2, int radius, bool tiled, void* stream,
                               bool half_in = false, const void* det = nullptr, bool det_half = false) {
  using namespace lgu;
  if (!means || !covs || !volume || !levels || L < 1 || L > VP_MAXL) return LGU_E_BADARG;
  if (E < 0 || H1 < 1 || W1 < 1 || H2 < 1 || W2 < 1 || radius < 0) return LGU_E_BADARG;
  for (int l = 0; l < L; l++)
    if (!levels[l]) return LGU_E_BADARG;
  if (W2 % 4 != 0 || ((reinterpret_cast<uintptr_t>(volume) | reinterpret_cast<uintptr_t>(levels[0])) & 15) != 0)
    return LGU_E_UNSUPPORTED;
  if (tiled && levels[0] == volume && (H2 % 4 != 0 || W2 % 8 != 0)) return LGU_E_BADARG;  // padded slice cannot alias
  if (half_in && static_cast<const void*>(levels[0]) == volume) return LGU_E_BADARG;        // nor can a half volume
  size_t floats = 0;
  for (int l = 0, h = H2, w = W2; l < L; l++, h >>= 1, w >>= 1) floats += (size_t)h * w;
  const size_t lds = floats * sizeof(float);
  if (lds > 96 * 1024 || (H2 >> (L - 1)) < 1 || (W2 >> (L - 1)) < 1) return LGU_E_UNSUPPORTED;
  if (E == 0) return LGU_OK;
  VolPyrParams p;
  p.means = means; p.covs = covs; p.vin = volume;
  p.det = det; p.det_half = det_half ? 1 : 0;
  for (int l = 0; l < VP_MAXL; l++) p.out[l] = l < L ? levels[l] : nullptr;
  p.npix = (size_t)E * H1 * W1; p.H2 = H2; p.W2 = W2; p.L = L; p.r = radius;
  auto kern = half_in ? (tiled ? volume_pyramid_kernel<true, true> : volume_pyramid_kernel<false, true>)
                      : (tiled ? volume_pyramid_kernel<true, false> : volume_pyramid_kernel<false, false>);
  if (half_in) { if (tiled) allow_max_dynamic_lds<&volume_pyramid_kernel<true, true>>(); else allow_max_dynamic_lds<&volume_pyramid_kernel<false, true>>(); }
  else { if (tiled) allow_max_dynamic_lds<&volume_pyramid_kernel<true, false>>(); else allow_max_dynamic_lds<&volume_pyramid_kernel<false, false>>(); }
  hipLaunchKernelGGL(kern, dim3((unsigned)p.npix), dim3(VP_THREADS), lds, reinterpret_cast<hipStream_t>(stream), p);
  return launch_status();
}

int lgu_volume_pyramid_f32(const float* means, const float* covs, const float* volume, float* const* levels, int L,
                           int E, int H1, int W1, int H2, int W2, int radius, void* stream) {
  return volume_pyramid_host(means, covs, volume, levels, L, E, H1, W1, H2, W2, radius, false, stream);
}

int lgu_volume_pyramid_tiled_f32(const float* means, const float* covs, const float* volume, float* const* levels,
                                 int L, int E, int H1, int W1, int H2, int W2, int radius, void* stream) {
  return volume_pyramid_host(means, covs, volume, levels, L, E, H1, W1, H2, W2, radius, true, stream);
}

int lgu_volume_pyramid_h16(const float* means, const float* covs, const void* volume, float* const* levels, int L,
                           int E, int H1, int W1, int H2, int W2, int radius, int tiled, void* stream) {
  return volume_pyramid_host(means, covs, volume, levels, L, E, H1, W1, H2, W2, radius, tiled != 0, stream, true);
}

int lgu_volume_pyramid_det(const float* means, const float* covs, const void* det, int det_half, const void* volume,
                           int volume_half, float* const* levels, int L, int E, int H1, int W1, int H2, int W2, int radius,
                           int tiled, void* stream) {
  if (!det) return LGU_E_BADARG;
  return volume_pyramid_host(means, covs, volume, levels, L, E, H1, W1, H2, W2, radius, tiled != 0, stream, volume_half != 0,
                             det, det_half != 0);
}

int lgu_volume_retile_f32(const float* src, float* dst, long long nslices, int H2, int W2, int to_tiled, void* stream) {
  using namespace lgu;
  if (!src || !dst || nslices < 0 || H2 < 1 || W2 < 1) return LGU_E_BADARG;
  if (nslices == 0) return LGU_OK;
  const size_t per = to_tiled ? (size_t)((H2 + 3) >> 2) * ((W2 + 7) >> 3) * 32 : (size_t)H2 * W2;
  const size_t want = ((size_t)nslices * per + 255) / 256;
  const unsigned grid = (unsigned)(want < 65535u * 32 ? want : 65535u * 32);
  hipLaunchKernelGGL(volume_retile_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst,
                     (size_t)nslices, H2, W2, to_tiled);
  return launch_status();
}

}  // extern "C"

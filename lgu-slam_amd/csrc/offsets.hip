// offsets.hip — post-processing of the learned sampling offsets in one pass (gfx950).
//
// Reference composition (droid_slam/modules/corr.py:117-135 for CorrBlock, :217-235 for AltCorrBlock, with
// per_Corr_Normalization of gaussianMask_cuda.py:26-33), per new edge set:
//     o0 = ofsMap(feats)                                        (E,C,h,w)   C = 2*rd*rd
//     o1 = interpolate(ofs_residual(avg_pool2d(feats, 2)), (h,w))           nearest
//     o0 = 4 * tanh((o0 - mean(o0)) / sqrt(var(o0) + eps))      statistics per sample over (C,h,w), biased variance
//     o1 = (4 * tanh((o1 - mean(o1)) / sqrt(var(o1) + eps)) + o0) / 2
//     offset[l] = o_l.permute(0,2,3,1) ... .float().contiguous()            what the samplers take: (E,h,w,rd,rd,2)
// = 4 reductions with one output per sample (latency-bound: ~40 us each at 20 edges), ~14 elementwise passes and two
// transposing copies.  Here: one statistics kernel (fp64 partial sums per chunk, no atomics: deterministic) and one
// finalize kernel that reads the two convolution outputs once (the low-resolution one through the nearest-neighbour
// index map, never materialising the upsampled tensor) and writes both offset tensors channel-last in fp32.
//
// Arithmetic mirrors the torch ops.  fp32 inputs: the same fp32 operations in the same order (x - m) / s, tanhf, * 4,
// (+ o0) / 2.  Half inputs (the convolutions run under autocast in factor_graph.add_factors): torch evaluates every op
// in fp32 and rounds its result to half, so each step below is rounded to half as well; autocast promotes the nearest
// upsampling to fp32, so there (is_half == 2) level 1 is fp32 arithmetic on the half-valued inputs plus the half-rounded
// level 0.  The statistics differ from torch's in summation order only (fp64 sums here).
#include "lgu_common.hpp"

namespace lgu {

constexpr int OF_CHUNKS = 64;    // partial sums per sample
constexpr int OF_THREADS = 256;
constexpr int OF_TP = 32;        // pixels per workgroup of the finalize kernel

struct OffParams {
  const void* o0;   // (E,C,H,W)
  const void* o1;   // (E,C,Hl,Wl)
  float* out0;      // (E,H,W,C)
  float* out1;
  double* partial;  // (E, OF_CHUNKS, 4): sum0, sumsq0, sum1, sumsq1
  int E, C, H, W, Hl, Wl;
  float sy, sx;     // Hl / H, Wl / W as torch's nearest interpolate forms them
  float eps;
  int l1_f32;       // half inputs under autocast: the upsampling is promoted to fp32 there, so level 1 is fp32 arithmetic
  const float* probe;  // optional (E, T, H, W): level 1 leaves scaled by sigmoid(var over the T probe samples) (corr.py:203-207)
  int T;
};

template <bool HALF>
__device__ __forceinline__ float of_load(const void* base, size_t i) {
  if (HALF) return (float)static_cast<const _Float16*>(base)[i];
  return static_cast<const float*>(base)[i];
}

// nearest-neighbour source index (ATen upsample_nearest: min(floor(dst * scale), in - 1))
__device__ __forceinline__ int of_src(int dst, float scale, int in) {
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

__device__ __forceinline__ float of_rh(float v) { return (float)(_Float16)v; }  // round to half, back to float

template <bool HALF>
__global__ __launch_bounds__(OF_THREADS) void offsets_stats_kernel(const OffParams p) {
  const int e = blockIdx.y, chunk = blockIdx.x;
  const int HW = p.H * p.W;
  const unsigned n = (unsigned)p.C * (unsigned)HW;              // (host-checked: C * H * W < 2^31)
  const unsigned per = (n + OF_CHUNKS - 1) / OF_CHUNKS;
  const unsigned lo = (unsigned)chunk * per, hi = lo + per < n ? lo + per : n;
  const size_t b0 = (size_t)e * n, b1 = (size_t)e * p.C * p.Hl * p.Wl;
  double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0;
  // eight elements per trip: their (unconditional, index-clamped) loads are issued together, then added in index order — the
  // sums are those of a one-at-a-time loop bit for bit; that loop paid a full load latency per element (a single edge is 64
  // workgroups for 1.9 MB)
  constexpr int UN = 8;
  for (unsigned i = lo + threadIdx.x; i < hi; i += OF_THREADS * UN) {
    float a[UN], b[UN];
    bool ok[UN];
#pragma unroll
    for (int j = 0; j < UN; j++) {
      const unsigned ij = i + (unsigned)j * OF_THREADS;
      ok[j] = ij < hi;
      const unsigned ic = ok[j] ? ij : hi - 1;
      const unsigned c = ic / (unsigned)HW, r = ic - c * (unsigned)HW;
      const int y = (int)(r / (unsigned)p.W), x = (int)(r - (unsigned)y * (unsigned)p.W);
      a[j] = of_load<HALF>(p.o0, b0 + ic);
      b[j] = of_load<HALF>(p.o1, b1 + ((size_t)c * p.Hl + of_src(y, p.sy, p.Hl)) * p.Wl + of_src(x, p.sx, p.Wl));
    }
#pragma unroll
    for (int j = 0; j < UN; j++)
      if (ok[j]) {
        const double da = a[j], db = b[j];
        s0 += da; q0 += da * da; s1 += db; q1 += db * db;
      }
  }
  __shared__ double red[OF_THREADS / kWave][4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    s0 += __shfl_xor(s0, m, kWave); q0 += __shfl_xor(q0, m, kWave);
    s1 += __shfl_xor(s1, m, kWave); q1 += __shfl_xor(q1, m, kWave);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wv][0] = s0; red[wv][1] = q0; red[wv][2] = s1; red[wv][3] = q1; }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0.0;
    for (int k = 0; k < OF_THREADS / kWave; k++) t += red[k][threadIdx.x];
    p.partial[((size_t)e * OF_CHUNKS + chunk) * 4 + threadIdx.x] = t;
  }
}

template <bool HALF>
__global__ __launch_bounds__(OF_THREADS) void offsets_finalize_kernel(const OffParams p) {
  extern __shared__ float of_sm[];  // [2][C][OF_TP + 1]
  __shared__ float stat[4];          // mean0, std0, mean1, std1
  __shared__ float mask[OF_TP];      // probe form: the uncertainty mask of the tile's pixels
  const int e = blockIdx.y;
  const int HW = p.H * p.W, C = p.C;
  // the partial sums arrive in LDS with one load per thread (read one after the other by the summing thread they cost a load
  // latency each), then two threads add them in chunk order: deterministic, and the same order as ever
  __shared__ double part[OF_CHUNKS * 4];
  static_assert(OF_CHUNKS * 4 == OF_THREADS, "one partial per thread");
  const double my_part = p.partial[(size_t)e * OF_CHUNKS * 4 + threadIdx.x];   // (in flight under the probe arithmetic below)
  const int pix0 = blockIdx.x * OF_TP;
  if (p.probe && threadIdx.x >= kWave && threadIdx.x < kWave + OF_TP) {  // the arithmetic of probe_mask_scale_kernel, on the second wave
    const int pp_ = threadIdx.x - kWave;
    float mk = 1.0f;
    if (pix0 + pp_ < HW) {
      const float* pr = p.probe + (size_t)e * p.T * HW + pix0 + pp_;
      float mean = 0.0f, ss = 0.0f;
      if (p.T <= 9) {   // the 3 x 3 probe: the samples are fetched together (two dependent passes over memory cost 18 load latencies)
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; t++) v[t] = pr[(size_t)(t < p.T ? t : p.T - 1) * HW];
#pragma unroll
        for (int t = 0; t < 9; t++)
          if (t < p.T) mean += v[t];
        mean /= (float)p.T;
#pragma unroll
        for (int t = 0; t < 9; t++)
          if (t < p.T) {
            const float d = v[t] - mean;
            ss += d * d;
          }
      } else {
        for (int t = 0; t < p.T; t++) mean += pr[(size_t)t * HW];
        mean /= (float)p.T;
        for (int t = 0; t < p.T; t++) {
          const float d = pr[(size_t)t * HW] - mean;
          ss += d * d;
        }
      }
      const float var = ss / (float)(p.T - 1);
      mk = 1.0f / (1.0f + expf(-var));
    }
    mask[pp_] = mk;
  }
  part[threadIdx.x] = my_part;
  __syncthreads();
  if (threadIdx.x < 2) {
    double s = 0.0, q = 0.0;
    for (int k = 0; k < OF_CHUNKS; k++) {
      s += part[k * 4 + threadIdx.x * 2 + 0];
      q += part[k * 4 + threadIdx.x * 2 + 1];
    }
    const double n = (double)C * HW;
    const double mean = s / n;
    double var = q / n - mean * mean;  // biased (unbiased=False)
    var = var > 0.0 ? var : 0.0;
    float m = (float)mean, sd;
    if (HALF && !(p.l1_f32 && threadIdx.x == 1)) {  // torch: mean / var come back as half tensors; var + eps and sqrt are half ops
      m = of_rh(m);
      sd = of_rh(sqrtf(of_rh(of_rh((float)var) + p.eps)));
    } else {
      sd = sqrtf((float)var + p.eps);
    }
    stat[threadIdx.x * 2 + 0] = m;
    stat[threadIdx.x * 2 + 1] = sd;
  }
  __syncthreads();
  const float m0 = stat[0], sd0 = stat[1], m1 = stat[2], sd1 = stat[3];
  float* const t0 = of_sm;
  float* const t1 = of_sm + (size_t)C * (OF_TP + 1);
  const size_t b0 = (size_t)e * C * HW, b1 = (size_t)e * C * p.Hl * p.Wl;
  // four elements per trip, their loads issued together (unconditional, index-clamped), as in the statistics kernel
  constexpr int UN = 4;
  for (int idx0 = threadIdx.x; idx0 < C * OF_TP; idx0 += OF_THREADS * UN) {
    float av[UN], bv[UN];
    int cc[UN], pq[UN];
    bool ok[UN];
#pragma unroll
    for (int j = 0; j < UN; j++) {
      const int idx = idx0 + j * OF_THREADS;
      const int ic = idx < C * OF_TP ? idx : C * OF_TP - 1;
      cc[j] = ic / OF_TP;
      pq[j] = ic - cc[j] * OF_TP;
      const int pix = pix0 + pq[j];
      ok[j] = idx < C * OF_TP && pix < HW;
      const int pxc = pix < HW ? pix : HW - 1;
      const int y = pxc / p.W, x = pxc - y * p.W;
      av[j] = of_load<HALF>(p.o0, b0 + (size_t)cc[j] * HW + pxc);
      bv[j] = of_load<HALF>(p.o1, b1 + ((size_t)cc[j] * p.Hl + of_src(y, p.sy, p.Hl)) * p.Wl + of_src(x, p.sx, p.Wl));
    }
#pragma unroll
    for (int j = 0; j < UN; j++) {
      if (!ok[j]) continue;
      const int c = cc[j], pp = pq[j];
      const float a = av[j], b = bv[j];
      float v0, v1;
      if (HALF) {
        v0 = of_rh(of_rh(tanhf(of_rh(of_rh(a - m0) / sd0))) * 4.0f);
        if (p.l1_f32) {
          v1 = (tanhf((b - m1) / sd1) * 4.0f + v0) / 2.0f;
        } else {
          const float u = of_rh(of_rh(tanhf(of_rh(of_rh(b - m1) / sd1))) * 4.0f);
          v1 = of_rh(of_rh(u + v0) / 2.0f);
        }
      } else {
        v0 = tanhf((a - m0) / sd0) * 4.0f;
        v1 = (tanhf((b - m1) / sd1) * 4.0f + v0) / 2.0f;
      }
      t0[c * (OF_TP + 1) + pp] = v0;
      t1[c * (OF_TP + 1) + pp] = v1;
    }
  }
  __syncthreads();
  const int npx = HW - pix0 < OF_TP ? HW - pix0 : OF_TP;
  float* const d0 = p.out0 + ((size_t)e * HW + pix0) * C;
  float* const d1 = p.out1 + ((size_t)e * HW + pix0) * C;
  for (int idx = threadIdx.x; idx < npx * C; idx += OF_THREADS) {  // channel-last rows: one contiguous run per tile
    const int pp = idx / C, c = idx - pp * C;
    d0[idx] = t0[c * (OF_TP + 1) + pp];
    const float v1 = t1[c * (OF_TP + 1) + pp];
    d1[idx] = p.probe ? v1 * mask[pp] : v1;   // = the stored value times the mask, as the separate in-place pass computes it
  }
}

// offset[1] *= sigmoid(var(probe)) of AltCorrBlock.corr_fn (reference corr.py:203-207): probe (E, T, H, W) holds the T = 9
// plain level-1 samples of every pixel, the variance is the unbiased one over them (torch.var default), the mask scales
// the pixel's C offset values.  One workgroup = 32 pixels: 32 threads form the masks, all threads scale 32 x C floats.
__global__ __launch_bounds__(OF_THREADS) void probe_mask_scale_kernel(const float* __restrict__ probe, float* __restrict__ offset,
                                                                      int HW, int T, int C) {
  __shared__ float mask[OF_TP];
  const int e = blockIdx.y, pix0 = blockIdx.x * OF_TP;
  if (threadIdx.x < OF_TP && pix0 + threadIdx.x < HW) {
    const float* pp = probe + (size_t)e * T * HW + pix0 + threadIdx.x;
    float mean = 0.0f;
    for (int t = 0; t < T; t++) mean += pp[(size_t)t * HW];
    mean /= (float)T;
    float ss = 0.0f;
    for (int t = 0; t < T; t++) {
      const float d = pp[(size_t)t * HW] - mean;
      ss += d * d;
    }
    const float var = ss / (float)(T - 1);
    mask[threadIdx.x] = 1.0f / (1.0f + expf(-var));
  }
  __syncthreads();
  const int npx = HW - pix0 < OF_TP ? HW - pix0 : OF_TP;
  float* const o = offset + ((size_t)e * HW + pix0) * C;
  for (int idx = threadIdx.x; idx < npx * C; idx += OF_THREADS) o[idx] *= mask[idx / C];
}

// Tail of GaussianMask.gaussian_parameters (reference droid_slam/gaussianMask_cuda.py:69-83) after the two 16 -> 2 linear
// heads: cov = sigmoid(per-sample standardised covMap output) * 5 + 0.05, det = cov.x * cov.y, mean = pixel grid +
// meanMap output — ~12 tiny launches (two of them one-output-per-sample reductions) in one: a workgroup per sample
// sums its H*W*2 values (fp64, fixed order), then finishes its pixels.  HALF: the heads ran under autocast; every step is
// rounded to half as the framework's half kernels do, cov is widened at the end (.float()), det stays half, and the mean
// is the fp32 sum of the fp32 grid and the half head output (type promotion) — bit for bit the torch composition unless
// a statistic lands across a half rounding boundary.
template <bool HALF>
__global__ __launch_bounds__(OF_THREADS) void ga_params_kernel(const void* __restrict__ mean_ofs, const void* __restrict__ cov_raw,
                                                               float* __restrict__ mean, float* __restrict__ cov,
                                                               void* __restrict__ det, int H, int W, float eps) {
  const int e = blockIdx.x, HW = H * W, n = HW * 2;
  const size_t base = (size_t)e * n;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < n; i += OF_THREADS) {
    const double v = of_load<HALF>(cov_raw, base + i);
    s += v; q += v * v;
  }
  __shared__ double red[OF_THREADS / kWave][2];
  __shared__ float stat[2];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { s += __shfl_xor(s, m, kWave); q += __shfl_xor(q, m, kWave); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
    for (int k = 0; k < OF_THREADS / kWave; k++) { ts += red[k][0]; tq += red[k][1]; }
    const double mu = ts / n;
    double var = tq / n - mu * mu;
    var = var > 0.0 ? var : 0.0;
    if (HALF) {
      stat[0] = of_rh((float)mu);
      stat[1] = of_rh(sqrtf(of_rh(of_rh((float)var) + eps)));
    } else {
      stat[0] = (float)mu;
      stat[1] = sqrtf((float)var + eps);
    }
  }
  __syncthreads();
  const float mu = stat[0], sd = stat[1];
  for (int pix = threadIdx.x; pix < HW; pix += OF_THREADS) {
    const size_t o = base + (size_t)pix * 2;
    float c[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const float x = of_load<HALF>(cov_raw, o + k);
      if (HALF) {
        const float z = of_rh(of_rh(x - mu) / sd);
        const float sg = of_rh(1.0f / (1.0f + expf(-z)));
        c[k] = of_rh(of_rh(sg * 5.0f) + 0.05f);
      } else {
        const float z = (x - mu) / sd;
        c[k] = 1.0f / (1.0f + expf(-z)) * 5.0f + 0.05f;
      }
      cov[o + k] = c[k];
    }
    if (HALF) static_cast<_Float16*>(det)[(size_t)e * HW + pix] = (_Float16)(c[0] * c[1]);
    else static_cast<float*>(det)[(size_t)e * HW + pix] = c[0] * c[1];
    const int y = pix / W, xg = pix - y * W;
    mean[o + 0] = (float)xg + of_load<HALF>(mean_ofs, o + 0);
    mean[o + 1] = (float)y + of_load<HALF>(mean_ofs, o + 1);
  }
}

}  // namespace lgu

extern "C" {

int lgu_gaussian_params(const void* mean_ofs, const void* cov_raw, float* mean, float* cov, void* det, int E, int H, int W,
                        int is_half, float eps, void* stream) {
  using namespace lgu;
  if (!mean_ofs || !cov_raw || !mean || !cov || !det || E < 0 || H < 1 || W < 1) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (is_half)
    hipLaunchKernelGGL(ga_params_kernel<true>, dim3(E), dim3(OF_THREADS), 0, st, mean_ofs, cov_raw, mean, cov, det, H, W, eps);
  else
    hipLaunchKernelGGL(ga_params_kernel<false>, dim3(E), dim3(OF_THREADS), 0, st, mean_ofs, cov_raw, mean, cov, det, H, W, eps);
  return launch_status();
}

int lgu_probe_mask_scale_f32(const float* probe, float* offset, int E, int HW, int T, int C, void* stream) {
  using namespace lgu;
  if (!probe || !offset || E < 0 || HW < 1 || T < 2 || C < 1) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  if (E > 65535) return LGU_E_UNSUPPORTED;
  hipLaunchKernelGGL(probe_mask_scale_kernel, dim3((HW + OF_TP - 1) / OF_TP, E), dim3(OF_THREADS), 0,
                     reinterpret_cast<hipStream_t>(stream), probe, offset, HW, T, C);
  return launch_status();
}

long long lgu_offsets_finalize_scratch_bytes(int E) {
  return E < 0 ? 0 : (long long)E * lgu::OF_CHUNKS * 4 * (long long)sizeof(double);
}

static int offsets_finalize_impl(const void* o0, const void* o1, const float* probe, int T, float* out0, float* out1, void* scratch,
                                 int E, int C, int H, int W, int Hl, int Wl, int is_half, float eps, void* stream);

int lgu_offsets_finalize(const void* o0, const void* o1, float* out0, float* out1, void* scratch, int E, int C, int H,
                         int W, int Hl, int Wl, int is_half, float eps, void* stream) {
  return offsets_finalize_impl(o0, o1, nullptr, 0, out0, out1, scratch, E, C, H, W, Hl, Wl, is_half, eps, stream);
}

/* lgu_offsets_finalize with the uncertainty mask of AltCorrBlock.corr_fn (reference corr.py:203-207) folded in: probe (E, T,
 * H, W) fp32 holds the T >= 2 plain level-1 samples per pixel; out1 = level-1 offsets * sigmoid(unbiased var over T) — bit for
 * bit what lgu_offsets_finalize followed by lgu_probe_mask_scale_f32 leaves there, without the extra pass over out1. */
int lgu_offsets_finalize_masked(const void* o0, const void* o1, const float* probe, int T, float* out0, float* out1,
                                void* scratch, int E, int C, int H, int W, int Hl, int Wl, int is_half, float eps, void* stream) {
  if (!probe || T < 2) return LGU_E_BADARG;
  return offsets_finalize_impl(o0, o1, probe, T, out0, out1, scratch, E, C, H, W, Hl, Wl, is_half, eps, stream);
}

static int offsets_finalize_impl(const void* o0, const void* o1, const float* probe, int T, float* out0, float* out1, void* scratch,
                                 int E, int C, int H, int W, int Hl, int Wl, int is_half, float eps, void* stream) {
  using namespace lgu;
  if (!o0 || !o1 || !out0 || !out1 || !scratch) return LGU_E_BADARG;
  if (E < 0 || C < 1 || H < 1 || W < 1 || Hl < 1 || Wl < 1) return LGU_E_BADARG;
  if (E == 0) return LGU_OK;
  const size_t lds = sizeof(float) * 2 * (size_t)C * (OF_TP + 1);
  if (lds > 60 * 1024 || E > 65535 || (size_t)C * H * W >= (1u << 31)) return LGU_E_UNSUPPORTED;
  OffParams p;
  p.o0 = o0; p.o1 = o1; p.out0 = out0; p.out1 = out1; p.partial = static_cast<double*>(scratch);
  p.E = E; p.C = C; p.H = H; p.W = W; p.Hl = Hl; p.Wl = Wl;
  p.sy = (float)Hl / (float)H; p.sx = (float)Wl / (float)W;
  p.eps = eps;
  p.l1_f32 = is_half == 2;
  p.probe = probe; p.T = T;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 g1(OF_CHUNKS, E), g2((H * W + OF_TP - 1) / OF_TP, E);
  if (is_half) {
    hipLaunchKernelGGL(offsets_stats_kernel<true>, g1, dim3(OF_THREADS), 0, st, p);
    hipLaunchKernelGGL(offsets_finalize_kernel<true>, g2, dim3(OF_THREADS), lds, st, p);
  } else {
    hipLaunchKernelGGL(offsets_stats_kernel<false>, g1, dim3(OF_THREADS), 0, st, p);
    hipLaunchKernelGGL(offsets_finalize_kernel<false>, g2, dim3(OF_THREADS), lds, st, p);
  }
  return launch_status();
}

}  // extern "C"
